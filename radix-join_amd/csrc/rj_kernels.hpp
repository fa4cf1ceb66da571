// rj_kernels.hpp — host-callable launchers of the gfx950 kernels (rj_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "rj_device.hpp"

namespace rj {

// Stream + optional per-kernel HIP-event timing.  Implemented in rj_context.hip.
// timed(): should this launch be timed?  If so it hands out the start/stop events the launch
// itself records (hipExtLaunchKernelGGL: the timestamps of the dispatch packet — no extra
// event packets, so no idle gaps at the kernel boundaries).
struct Launch {
    hipStream_t stream;
    bool (*timed)(void* self, const char* name, hipEvent_t* start, hipEvent_t* stop);
    void* self;
};

// A launch the runtime rejected, or a template combination no kernel exists for: throws
// rj::Error (RJ_ERR_DEVICE / RJ_ERR_UNSUPPORTED).  Implemented in rj_context.hip.
[[noreturn]] void launch_failed(const char* kernel, const char* what, bool unsupported);

// ---- page metadata / decode (replaces Table::from_columnar, reference
//      src/build_table.cpp:312-436, for fixed-width columns)
// flags[0] = #pages breaking the "regular" shape, flags[1..2] = total rows (u64)
void launch_page_headers(const Launch& L, const uint8_t* pages, uint32_t n_pages, uint32_t rows_full,
                         uint32_t* page_rows, unsigned long long* flags);
// flag[0] += pages that hold a NON-NULL value at a row index >= num_rows (the reference's
// "row_idx" error, src/build_table.cpp:334-336); row_base = exclusive scan of the page rows
void launch_rows_beyond(const Launch& L, const uint8_t* pages, uint32_t n_pages,
                        const uint32_t* row_base, uint64_t num_rows, unsigned long long* flag);
void launch_decode_pages(const Launch& L, const uint8_t* pages, uint32_t n_pages, int width,
                         const uint32_t* row_base, uint64_t num_rows, uint8_t* values,
                         uint8_t* valid);

// ---- small single-workgroup scans
// off[i] = sum_{j<i} in[j], off[n] = total; cursor (optional) = copy of off[0..n)
void launch_scan_bins(const Launch& L, const uint32_t* in, uint32_t n, uint32_t* off,
                      uint32_t* cursor);
// per-segment scan: off[seg*F+d] = seg_off[seg] + prefix of hist[seg][0..d); off[nseg*F] = total
void launch_scan_segments(const Launch& L, const uint32_t* hist, const uint32_t* seg_off,
                          uint32_t nseg, uint32_t F, uint32_t xcd_log2, uint32_t* off, uint32_t* cursor);
// grp_start[s] = sum_{t<s} ceil(len_t / group_tuples) over segments (seg_off[nseg+1]; with seg_end,
// segment t is [seg_off[t], seg_end[t]) and seg_off needs only nseg entries)
void launch_group_table(const Launch& L, const uint32_t* seg_off, uint32_t nseg,
                        uint32_t group_tuples, uint32_t* grp_start, const uint32_t* seg_end = nullptr);

// ---- radix partition pass (replaces the ≤128-way hash partition of row indices,
//      reference src/execute.cpp:124-184)
// Source pass: tuples are formed from columns (page decode fused in).
void launch_pass_hist_src(const Launch& L, const TupleSrc& src, int key_words, const PassParams& pp,
                          uint32_t n_groups);
// Fine histogram of a two-pass plan (both digits in one read of the source): fine[d1 * F2 + d2],
// pre-zeroed; `grid` persistent workgroups (one per CU).  launch_scan_fine turns it into the
// pass-2 offsets/cursors (off2[F1*F2 + 1], cursor2) and the pass-1 ones (off1[F1 + 1], cursor1).
// xcd_tpg > 0 (grid a multiple of 8): XCD-aware placement of the first pass — workgroup w counts
// the tile groups (of xcd_tpg tiles) g with g & 7 == w & 7 and also adds its pass-1 digit counts
// to coarse_x[d1 * 8 + (w & 7)] (pre-zeroed); launch_scan_fine then writes cursor1[d1 * 8 + x]
void launch_fine_hist_src(const Launch& L, const TupleSrc& src, int key_words, uint32_t shift,
                          uint32_t b1, uint32_t b2, uint32_t grid, uint32_t* fine, uint32_t xcd_tpg = 0,
                          uint32_t* coarse_x = nullptr);
// the same over tuples that already sit in the partition layout (hashed words / packed pairs):
// the passes behind a sharded join's exchange
void launch_fine_hist_words(const Launch& L, const Words& in, bool packed, uint32_t n, uint32_t shift,
                            uint32_t b1, uint32_t b2, uint32_t grid, uint32_t* fine, uint32_t xcd_tpg = 0,
                            uint32_t* coarse_x = nullptr);
void launch_scan_fine(const Launch& L, const uint32_t* fine, uint32_t F1, uint32_t F2,
                      uint32_t* off2, uint32_t* cursor2, uint32_t* off1, uint32_t* cursor1,
                      const uint32_t* coarse_x = nullptr);
// aos3 (key_words == 1, carry_words == 2, last pass of a plan): out.w[0] receives 12-byte
// {hashed key, carry lo, carry hi} tuples instead of a key array + a pair array
void launch_pass_scatter_src(const Launch& L, const TupleSrc& src, int key_words, int carry_words,
                             const PassParams& pp, uint32_t n_groups, const Words& out, bool aos3 = false);
// Dense pass: re-partition every segment of already partitioned word arrays.
void launch_pass_hist_dense(const Launch& L, const Words& in, const PassParams& pp,
                            uint32_t n_groups);
// pair_word >= 0: words pair_word / pair_word+1 (a two-word carry) live as 8-byte pairs in
// in.w[pair_word] / out.w[pair_word] (what every pass writes for two-word carries); else -1
void launch_pass_scatter_dense(const Launch& L, const Words& in, int n_words, int pair_word,
                               const PassParams& pp, uint32_t n_groups, const Words& out, bool aos3 = false);

// 12-byte tuples between the passes of a key + two-word-carry plan: the histogram of a later
// pass reads the 16-bit digit side array its predecessor wrote (PassParams::side_out), the
// scatter reads and writes 12-byte tuples
void launch_pass_hist_digits(const Launch& L, const uint16_t* digits, const PassParams& pp, uint32_t n_groups);
// ... or, without a side array, fishes the keys out of the 12-byte tuples themselves
void launch_pass_hist_aos3(const Launch& L, const uint32_t* in_tuples, const PassParams& pp, uint32_t n_groups);
void launch_pass_scatter_aos3(const Launch& L, const uint32_t* in_tuples, const PassParams& pp, uint32_t n_groups,
                              uint32_t* out_tuples);

// Packed layout ({hashed key, carry} pairs in one array) for one key word + one carry word.
void launch_pass_hist_packed(const Launch& L, const uint32_t* in_pairs, const PassParams& pp,
                             uint32_t n_groups);
void launch_pass_scatter_src_packed(const Launch& L, const TupleSrc& src, const PassParams& pp,
                                    uint32_t n_groups, uint32_t* out_pairs, bool blocked_out = false);
// (pairs between the passes of a packed plan in blocks of 16 keys + 16 carries: the next histogram reads the keys only)
void launch_pass_hist_blocked(const Launch& L, const uint32_t* in_blocked, const PassParams& pp, uint32_t n_groups);
void launch_pass_scatter_blocked(const Launch& L, const uint32_t* in_blocked, const PassParams& pp, uint32_t n_groups,
                                 uint32_t* out_pairs, bool blocked_out);
void launch_pass_scatter_packed(const Launch& L, const uint32_t* in_pairs, const PassParams& pp,
                                uint32_t n_groups, uint32_t* out_pairs);

// ---- build + probe (replaces reference src/execute.cpp:196-249)
void launch_heavy_tasks(const Launch& L, const uint32_t* offR, const uint32_t* offS, uint32_t NP,
                        uint32_t* tasks, uint32_t* n_heavy, uint32_t max_tasks);
// grid = jp.heavy_grid + ceil(jp.NP / join_partitions_per_workgroup(...))
uint32_t join_partitions_per_workgroup(int key_words, int cw_build, const JoinParams& jp);
void launch_join(const Launch& L, int key_words, int cw_build, int cw_probe, const JoinParams& jp,
                 uint32_t grid);

// Broadcast join: build side of at most JN_RMAX rows, no partitioning; `grid` workgroups stride
// over the probe rows in chunks of JN_SUB.
void launch_join_bcast(const Launch& L, int key_words, int cw_build, int cw_probe,
                       const BcastParams& bp, uint32_t grid);

// ---- materialise (replaces the per-row output copy, reference src/execute.cpp:236-242,
//      and Table::to_columnar, src/build_table.cpp:456-594)
void launch_gather(const Launch& L, const ColRef& src, const uint32_t* idx, uint64_t n,
                   const OutStream& dst, uint8_t* dst_valid);
// test hook: keeps the stream busy for `ms` milliseconds (bounded; see RJ_DEBUG_SHARD_FAIL)
void launch_debug_stall(const Launch& L, uint32_t ms);
// wide carries (several payload columns travelling with the key): validity bits of up to three
// columns as one word per row; the emitted records -> one dense array (+ validity bytes) per column
void launch_pack_validity(const Launch& L, const uint8_t* v0, const uint8_t* v1, const uint8_t* v2, uint32_t n,
                          uint32_t* out);
void launch_split_records(const Launch& L, const SplitParams& sp, uint64_t n);
void launch_finish_pages(const Launch& L, uint8_t* pages, uint64_t n_rows, int width);
// headers + bitmaps of up to three probe-written streams, row count read on the device
void launch_finish_streams(const Launch& L, uint8_t* const* pages, const int* widths, uint32_t n,
                           const unsigned long long* n_rows_dev, uint64_t cap_rows);
void launch_encode_nullable(const Launch& L, const uint8_t* values, const uint8_t* valid,
                            uint64_t n_rows, int width, uint8_t* pages);

// ---- VARCHAR gather + page encode on the device (replaces, for large results, the string half
//      of Table::to_columnar, reference src/build_table.cpp:595-677)
void launch_vc_resolve(const Launch& L, const uint8_t* pages, uint32_t n_pages, const uint32_t* row_base,
                       const uint32_t* rowids, uint32_t n, VcRow* out);
// VARCHAR join keys: 64-bit FNV-1a of every row's string (+ where it sits, + validity), the
// byte-for-byte check of the joined pairs, and the compaction that only a hash collision triggers
void launch_vc_hash(const Launch& L, const uint8_t* pages, uint32_t n_pages, const uint32_t* row_base,
                    const uint32_t* rowids, uint32_t n, VcRow* rows, uint64_t* hash, uint8_t* valid,
                    uint64_t hash_mask = ~0ull);
void launch_vc_verify(const Launch& L, const uint8_t* pages_b, uint32_t np_b, const VcRow* rows_b,
                      const uint8_t* pages_p, uint32_t np_p, const VcRow* rows_p, const uint32_t* bidx,
                      const uint32_t* pidx, uint32_t n, uint8_t* keep, unsigned long long* n_bad);
void launch_vc_compact(const Launch& L, const uint8_t* keep, const uint32_t* bidx, const uint32_t* pidx, uint32_t n,
                       uint32_t* out_b, uint32_t* out_p, unsigned long long* cursor);
// page_out == nullptr: pages_in_chunk[c] = pages of chunk c; else the pages are written to
// page_out[page_base[c] ...]
void prewarm_varchar_dev(const Launch& L, uint32_t* zeroed);  // (RJ_CTX_PREWARM: loads the code object)
void launch_vc_walk(const Launch& L, const VcRow* rows, uint32_t n, uint32_t* pages_in_chunk,
                    const uint32_t* page_base, VcPage* page_out);
void launch_vc_encode(const Launch& L, const uint8_t* pages, uint32_t n_pages, const VcRow* rows,
                      const VcPage* plist, uint32_t n_out_pages, uint8_t* out);

}  // namespace rj
