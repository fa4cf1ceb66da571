// rj_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the radix join.
//
// Everything here is integer / indexing work bound by HBM bandwidth (no MFMA):
//   K0  page headers        4 B read per 8 KiB page
//   K1  page decode         irregular (NULL-bearing) pages -> dense values + validity
//   K2  pass histogram      per-workgroup LDS histogram, flushed once per tile group
//   K3  bin scan            one workgroup per input segment, wave-shuffle scan
//   K4  pass scatter        tile of 16384 tuples sorted by digit in LDS (LDS atomics give the
//                           rank), runs written out contiguously = software write-combining
//   K5/6 build + probe      per-partition bucketised hash table in LDS (one 16-byte read per
//                           probe), ballot/mbcnt output offsets, one global reservation per
//                           4096 probe tuples, streams written straight into Page images
//   K7  gather              row-id -> column value (generic late materialisation)
//   K8  page finish/encode  Page headers + validity bitmaps of the result
//
// Reference counterparts are cited at each kernel (paths relative to the reference).
#include <hip/hip_ext.h>

#include "rj_kernels.hpp"

#include <algorithm>
#include <type_traits>

namespace rj {

// ============================================================== small helpers
// Bijective finalisers (murmur3 fmix32 / fmix64).  The reference hashes with
// fmix64 (src/execute.cpp:21-27); the hash is not observable in results
// (SURVEY.md §2 #11), so INT32 keys use the cheaper 32-bit mix.  Because both
// are bijections the partitions store HASHED keys and the probe un-hashes on
// emit; equal hashed keys <=> equal keys.
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t unfmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x7ed1b41du;  // inverse of 0xc2b2ae35 mod 2^32
    h ^= (h >> 13) ^ (h >> 26);
    h *= 0xa5cb9243u;  // inverse of 0x85ebca6b mod 2^32
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint64_t fmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}
__device__ __forceinline__ uint64_t unfmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0x9cb4b2f8129337dbULL;  // inverse of 0xc4ceb9fe1a85ec53 mod 2^64
    k ^= k >> 33;
    k *= 0x4f74430c22a54005ULL;  // inverse of 0xff51afd7ed558ccd mod 2^64
    k ^= k >> 33;
    return k;
}

// lanes below me whose bit is set in `mask` (wave64)
__device__ __forceinline__ uint32_t lane_prefix(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for
// vmcnt(0), i.e. for every outstanding global load AND store of the wave; the kernels
// below never hand global data from wave to wave inside a launch, so their barriers only
// need the LDS counter drained — global stores then retire asynchronously behind the
// next phase instead of stalling every barrier.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 16-byte vector of four tuples' words.  Tuple runs start at arbitrary (4-byte aligned)
// offsets; gfx950 global loads only need dword alignment, so one dwordx4 replaces four
// dword loads and their address arithmetic.
typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));

// Exclusive scan of one value per thread across the workgroup (blockDim.x a
// multiple of 64, at most 1024).  s_wsum needs blockDim.x/64 words.  Contains
// one barrier; the caller must sync again before reusing s_wsum.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_wsum, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    uint32_t       incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(incl, off);
        if (lane >= (uint32_t)off) incl += t;
    }
    if (lane == 63) s_wsum[wid] = incl;
    lds_barrier();
    uint32_t wbase = 0, tot = 0;
    for (uint32_t k = 0; k < nw; ++k) {
        uint32_t s = s_wsum[k];
        if (k < wid) wbase += s;
        tot += s;
    }
    total = tot;
    return wbase + incl - v;
}

__device__ __forceinline__ uint32_t col_load32(const ColRef& c, uint32_t row) {
    if (c.kind == COL_PAGED) {
        uint32_t p = row / ROWS32, i = row - p * ROWS32;
        return *reinterpret_cast<const uint32_t*>(c.ptr + (size_t)p * PAGE_BYTES + HDR32 + i * 4u);
    } else if (c.kind == COL_DENSE) {
        return reinterpret_cast<const uint32_t*>(c.ptr)[row];
    }
    return row;  // COL_IOTA
}
__device__ __forceinline__ uint64_t col_load64(const ColRef& c, uint32_t row) {
    if (c.kind == COL_PAGED) {
        uint32_t p = row / ROWS64, i = row - p * ROWS64;
        return *reinterpret_cast<const uint64_t*>(c.ptr + (size_t)p * PAGE_BYTES + HDR64 + i * 8u);
    } else if (c.kind == COL_DENSE) {
        return reinterpret_cast<const uint64_t*>(c.ptr)[row];
    }
    return row;
}

__device__ __forceinline__ void stream_store(const OutStream& o, uint64_t row, uint32_t lo,
                                             uint32_t hi, uint32_t x2 = 0u) {
    switch (o.mode) {
    case ST_DENSE96: {
        uint32_t* r = reinterpret_cast<uint32_t*>(o.base) + row * 3u;
        r[0] = lo;
        r[1] = hi;
        r[2] = x2;
        break;
    }
    case ST_DENSE32: reinterpret_cast<uint32_t*>(o.base)[row] = lo; break;
    case ST_DENSE64:
        reinterpret_cast<uint64_t*>(o.base)[row] = (uint64_t)lo | ((uint64_t)hi << 32);
        break;
    case ST_PAGED32: {
        uint32_t r = (uint32_t)row, p = r / ROWS32, i = r - p * ROWS32;
        *reinterpret_cast<uint32_t*>(o.base + (size_t)p * PAGE_BYTES + HDR32 + i * 4u) = lo;
        break;
    }
    case ST_PAGED64: {
        uint32_t r = (uint32_t)row, p = r / ROWS64, i = r - p * ROWS64;
        *reinterpret_cast<uint64_t*>(o.base + (size_t)p * PAGE_BYTES + HDR64 + i * 8u) =
            (uint64_t)lo | ((uint64_t)hi << 32);
        break;
    }
    default: break;
    }
}

// ================================================================= K0 headers
// One wave per page: reads `u16 n_rows @0` and the validity bitmap in the last
// (n_rows+7)/8 bytes (layout: reference src/build_table.cpp:326-331).  A column is
// "regular" when every page but the last holds exactly rows_full rows and every bitmap bit
// of every page is set; regular columns are addressed in place by every later kernel, the
// others go through K1 once.  The reference decodes fixed-width pages from the bitmap alone
// and never reads the header's non-null count (:332-343), so neither does this.
__global__ __launch_bounds__(256) void k_page_headers(const uint8_t* pages, uint32_t n_pages,
                                                      uint32_t rows_full, uint32_t* page_rows,
                                                      unsigned long long* flags) {
    // every workgroup walks a strip of pages (one wave per page at a time) and adds its two
    // sums with ONE pair of global atomics: a million pages must not queue on two words
    __shared__ unsigned long long s_irr[4], s_rows[4];
    const uint32_t     lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    unsigned long long irr = 0, rows = 0;
    for (uint32_t p = blockIdx.x * 4u + wid; p < n_pages; p += gridDim.x * 4u) {
        const uint8_t* page = pages + (size_t)p * PAGE_BYTES;
        const uint32_t nr = *reinterpret_cast<const uint16_t*>(page);
        const uint32_t nb = (nr + 7) / 8;
        const uint8_t* bm = page + PAGE_BYTES - nb;
        bool           ones = true;
        // full pages: 248 bitmap bytes at a dword boundary (INT32) or 126 at a halfword boundary
        // (INT64/FP64) — one load per lane; anything else byte by byte
        const uint32_t boff = PAGE_BYTES - nb;
        if ((nr & 31u) == 0 && (boff & 3u) == 0) {
            for (uint32_t k = lane; k < nb / 4; k += 64)
                ones = ones && reinterpret_cast<const uint32_t*>(bm)[k] == 0xffffffffu;
        } else if ((boff & 1u) == 0 && (nb & 1u) == 0) {
            for (uint32_t k = lane; k < nb / 2; k += 64) {
                const uint32_t rem = nr - k * 16u;
                const uint32_t want = rem >= 16 ? 0xffffu : ((1u << rem) - 1u);
                ones = ones && ((uint32_t)reinterpret_cast<const uint16_t*>(bm)[k] & want) == want;
            }
        } else {
            for (uint32_t k = lane; k < nb; k += 64) {
                const uint32_t rem = nr - k * 8u;
                const uint32_t want = rem >= 8 ? 0xffu : ((1u << rem) - 1u);
                ones = ones && ((bm[k] & want) == want);
            }
        }
        const bool all_ones = __ballot(!ones) == 0;
        if (lane == 0) {
            page_rows[p] = nr;
            irr += !all_ones || (p + 1 < n_pages ? nr != rows_full : (nr > rows_full || nr == 0));
            rows += nr;
        }
    }
    if (lane == 0) {
        s_irr[wid] = irr;
        s_rows[wid] = rows;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long i = s_irr[0] + s_irr[1] + s_irr[2] + s_irr[3];
        const unsigned long long r = s_rows[0] + s_rows[1] + s_rows[2] + s_rows[3];
        if (i) atomicAdd(&flags[0], i);
        if (r) atomicAdd(&flags[1], r);
    }
}

// "row_idx" rule of Table::from_columnar (reference src/build_table.cpp:334-336): the
// reference throws only when a NON-NULL value lands at a row index >= num_rows; NULL rows
// past the end are tolerated.  One wave per page; flag[0] counts offending pages.
__global__ __launch_bounds__(256) void k_rows_beyond(const uint8_t* pages, uint32_t n_pages,
                                                     const uint32_t* row_base, uint64_t num_rows,
                                                     unsigned long long* flag) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t p = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= n_pages) return;
    const uint8_t* page = pages + (size_t)p * PAGE_BYTES;
    const uint32_t nr = *reinterpret_cast<const uint16_t*>(page);
    const uint64_t rb = row_base[p];
    if (rb + nr <= num_rows) return;
    const uint8_t* bm = page + PAGE_BYTES - (nr + 7) / 8;
    bool           bad = false;
    for (uint32_t i = lane; i < nr; i += 64)
        if (rb + i >= num_rows && ((bm[i >> 3] >> (i & 7u)) & 1u)) bad = true;
    if (__ballot(bad) != 0 && lane == 0) atomicAdd(flag, 1ull);
}

// ================================================================== K1 decode
// One workgroup per page: rows -> dense values + validity bytes.  The value
// index of row i is the number of set validity bits below i (values are stored
// densely, the bitmap sits in the last (n_rows+7)/8 bytes: reference
// src/build_table.cpp:325-381).  Per 64 rows one ballot + mbcnt.
template <int WIDTH>
__global__ __launch_bounds__(256) void k_decode_pages(const uint8_t* pages, const uint32_t* row_base,
                                                      uint64_t num_rows, uint8_t* values,
                                                      uint8_t* valid) {
    __shared__ uint32_t s_w[4];
    const uint8_t*      page = pages + (size_t)blockIdx.x * PAGE_BYTES;
    const uint32_t      nr = *reinterpret_cast<const uint16_t*>(page);
    const uint32_t      rb = row_base[blockIdx.x];
    const uint8_t*      bitmap = page + PAGE_BYTES - (nr + 7) / 8;
    const uint8_t*      vals = page + (WIDTH == 4 ? HDR32 : HDR64);
    const uint32_t      lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    uint32_t            running = 0;
    for (uint32_t base = 0; base < nr; base += 256) {
        uint32_t i = base + threadIdx.x;
        bool     bit = i < nr ? ((bitmap[i >> 3] >> (i & 7u)) & 1u) : false;
        uint64_t mask = __ballot(bit);
        uint32_t pre = lane_prefix(mask);
        if (lane == 0) s_w[wid] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t wpre = 0, tot = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            uint32_t c = s_w[k];
            if (k < wid) wpre += c;
            tot += c;
        }
        uint64_t row = (uint64_t)rb + i;
        if (i < nr && row < num_rows) {
            valid[row] = bit ? 1 : 0;
            uint32_t vi = running + wpre + pre;
            if (WIDTH == 4) {
                reinterpret_cast<uint32_t*>(values)[row] =
                    bit ? reinterpret_cast<const uint32_t*>(vals)[vi] : 0u;
            } else {
                reinterpret_cast<uint64_t*>(values)[row] =
                    bit ? reinterpret_cast<const uint64_t*>(vals)[vi] : 0ull;
            }
        }
        running += tot;
        __syncthreads();
    }
}

// ==================================================================== K3 scan
// Single workgroup of 1024 threads; thread t owns a contiguous chunk.
// MODE 0: v[i] = in[i].  MODE 1: v[i] = ceil((in[i+1]-in[i]) / div)  (group table)
// (in_end, MODE 1 only: segment i ends at in_end[i] instead of in[i + 1])
template <int MODE>
__global__ __launch_bounds__(1024) void k_scan_bins(const uint32_t* in, const uint32_t* in_end, uint32_t n,
                                                    uint32_t div, uint32_t* off, uint32_t* cursor) {
    __shared__ uint32_t s_wsum[16];
    const uint32_t      chunk = (n + 1023u) / 1024u;
    const uint32_t      b = threadIdx.x * chunk;
    const uint32_t      e = min(n, b + chunk);
    uint32_t            sum = 0;
    for (uint32_t i = b; i < e; ++i) {
        uint32_t v = MODE == 0 ? in[i] : ((in_end ? in_end[i] : in[i + 1]) - in[i] + div - 1u) / div;
        sum += v;
    }
    uint32_t total;
    uint32_t run = block_excl_scan(sum, s_wsum, total);
    for (uint32_t i = b; i < e; ++i) {
        uint32_t v = MODE == 0 ? in[i] : ((in_end ? in_end[i] : in[i + 1]) - in[i] + div - 1u) / div;
        off[i] = run;
        if (cursor) cursor[i] = run;
        run += v;
    }
    if (threadIdx.x == 0) off[n] = total;
}

// Exclusive scan of the bins of ONE input segment per workgroup.  A segment's tuples are
// exactly the sum of its bins (NULL keys were dropped in the first pass), so the global
// offset of bin (seg, d) is seg_off[seg] + prefix inside the segment: every segment scans
// independently instead of one workgroup walking all nseg*F bins.
__global__ __launch_bounds__(PT_MAXF) void k_scan_segments(const uint32_t* hist,
                                                           const uint32_t* seg_off, uint32_t nseg,
                                                           uint32_t F, uint32_t xcd_log2, uint32_t* off,
                                                           uint32_t* cursor) {
    __shared__ uint32_t s_wsum[PT_MAXF / 64];
    const uint32_t      seg = blockIdx.x, d = threadIdx.x, X = 1u << xcd_log2;
    // a partition's sub-ranges (one per XCD, see PassParams::xcd_log2) lie one behind the other
    const size_t        b0 = ((size_t)seg * F + d) << xcd_log2;
    uint32_t            v = 0;
    if (d < F)
        for (uint32_t x = 0; x < X; ++x) v += hist[b0 + x];
    uint32_t            total;
    const uint32_t      ex = block_excl_scan(v, s_wsum, total);
    const uint32_t      base = seg_off ? seg_off[seg] : 0u;
    if (d < F) {
        off[(size_t)seg * F + d] = base + ex;
        uint32_t at = base + ex;
        for (uint32_t x = 0; x < X; ++x) {
            cursor[b0 + x] = at;
            at += hist[b0 + x];
        }
    }
    if (seg + 1 == nseg && d == 0) off[(size_t)nseg * F] = base + total;
}

// ========================================================== tuple loaders (K2/K4)
// Loader concept:  key(i, hk) -> valid      load<NW>(i, w) -> valid
// A loader fills one tile: item j of thread t is element base + j*PT_THREADS + t.  Every
// load is UNCONDITIONAL (the index is clamped to the last valid element) and all uniform
// decisions (column kind, validity array present, carry mode) are taken once per tile, so
// the compiler can issue the whole tile's loads back to back behind a single wait.  The
// returned bit mask says which items hold a tuple.
// PAIR >= 0: in.w[PAIR] is an array of 8-byte {word PAIR, word PAIR+1} pairs (a two-word carry)
// (issue_tile / finish_tile: the two halves of load_tile for the scatter's software pipeline —
// issue only ISSUES the tile's loads, finish does what needs the data.  Loaders of tuples that are
// already in the partition layout have nothing to finish.)
#define RJ_TRIVIAL_ISSUE_FINISH                                                                              \
    template <int NW>                                                                                        \
    __device__ __forceinline__ uint32_t issue_tile(uint32_t base, uint32_t end, uint32_t (&w)[PT_ITEMS][NW]) const { \
        return this->template load_tile<NW>(base, end, w);                                                   \
    }                                                                                                        \
    template <int NW>                                                                                        \
    __device__ __forceinline__ uint32_t finish_tile(uint32_t, uint32_t, uint32_t ok, uint32_t (&)[PT_ITEMS][NW]) const { \
        return ok;                                                                                           \
    }

template <int PAIR>
struct DenseLoaderT {
    static constexpr int pair = PAIR;
    Words in;
    RJ_TRIVIAL_ISSUE_FINISH
    // two-halves interface of the fine histogram (see SrcLoader): the words are hashed already
    __device__ __forceinline__ uint32_t issue_keys(uint32_t base, uint32_t end, uint32_t (&lo)[PT_ITEMS],
                                                   uint32_t (&)[PT_ITEMS]) const {
        return key_tile(base, end, lo);
    }
    __device__ __forceinline__ uint32_t finish_keys(uint32_t, uint32_t, uint32_t ok, uint32_t (&)[PT_ITEMS],
                                                    uint32_t (&)[PT_ITEMS]) const {
        return ok;
    }
    __device__ __forceinline__ uint32_t key_tile(uint32_t base, uint32_t end,
                                                 uint32_t (&hk)[PT_ITEMS]) const {
        if (base + PT_TILE <= end) {  // full tile: four consecutive tuples per 16-byte load
#pragma unroll
            for (int v = 0; v < PT_ITEMS / 4; ++v) {
                u32x4a x = *reinterpret_cast<const u32x4a*>(in.w[0] + base +
                                                            (v * PT_THREADS + threadIdx.x) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) hk[4 * v + e] = x[e];
            }
            return PT_ALL_ITEMS;
        }
        uint32_t ok = 0;
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
            uint32_t i = base + j * PT_THREADS + threadIdx.x;
            hk[j] = in.w[0][min(i, end - 1u)];
            ok |= (uint32_t)(i < end) << j;
        }
        return ok;
    }
    template <int NW>
    __device__ __forceinline__ uint32_t load_tile(uint32_t base, uint32_t end,
                                                  uint32_t (&w)[PT_ITEMS][NW]) const {
        if (base + PT_TILE <= end) {
#pragma unroll
            for (int v = 0; v < PT_ITEMS / 4; ++v) {
                const uint32_t i0 = base + (v * PT_THREADS + threadIdx.x) * 4;
#pragma unroll
                for (int a = 0; a < NW; ++a) {
                    if (a == pair) {  // four consecutive pairs = two 16-byte loads
                        const uint32_t* pp2 = in.w[a] + (size_t)i0 * 2;
                        u32x4a x = *reinterpret_cast<const u32x4a*>(pp2);
                        u32x4a y = *reinterpret_cast<const u32x4a*>(pp2 + 4);
                        if constexpr (NW >= 2) {
                            w[4 * v + 0][a] = x[0];
                            w[4 * v + 0][a + 1 < NW ? a + 1 : a] = x[1];
                            w[4 * v + 1][a] = x[2];
                            w[4 * v + 1][a + 1 < NW ? a + 1 : a] = x[3];
                            w[4 * v + 2][a] = y[0];
                            w[4 * v + 2][a + 1 < NW ? a + 1 : a] = y[1];
                            w[4 * v + 3][a] = y[2];
                            w[4 * v + 3][a + 1 < NW ? a + 1 : a] = y[3];
                        }
                    } else if (a != pair + 1 || pair < 0) {
                        u32x4a x = *reinterpret_cast<const u32x4a*>(in.w[a] + i0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) w[4 * v + e][a] = x[e];
                    }
                }
            }
            return PT_ALL_ITEMS;
        }
        uint32_t ok = 0;
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
            uint32_t i = base + j * PT_THREADS + threadIdx.x;
            uint32_t ic = min(i, end - 1u);
#pragma unroll
            for (int a = 0; a < NW; ++a) {
                if (a == pair) {
                    const uint2 t = reinterpret_cast<const uint2*>(in.w[a])[ic];
                    w[j][a] = t.x;
                    w[j][a + 1 < NW ? a + 1 : a] = t.y;
                } else if (a != pair + 1 || pair < 0) {
                    w[j][a] = in.w[a][ic];
                }
            }
            ok |= (uint32_t)(i < end) << j;
        }
        return ok;
    }
};

using DenseLoader = DenseLoaderT<-1>;


// One array of 12-byte {hashed key, carry lo, carry hi} tuples (what every pass of a key +
// two-word-carry plan writes): four consecutive tuples = three 16-byte loads.
struct Aos3Loader {
    const uint32_t* in;
    RJ_TRIVIAL_ISSUE_FINISH
    template <int NW>
    __device__ __forceinline__ uint32_t load_tile(uint32_t base, uint32_t end, uint32_t (&w)[PT_ITEMS][NW]) const {
        static_assert(NW == 3, "12-byte tuples");
        if (base + PT_TILE <= end) {
#pragma unroll
            for (int v = 0; v < PT_ITEMS / 4; ++v) {
                const uint32_t* p = in + (size_t)(base + (v * PT_THREADS + threadIdx.x) * 4) * 3;
                const u32x4a    x = *reinterpret_cast<const u32x4a*>(p), y = *reinterpret_cast<const u32x4a*>(p + 4),
                             z = *reinterpret_cast<const u32x4a*>(p + 8);
                w[4 * v + 0][0] = x[0];
                w[4 * v + 0][1] = x[1];
                w[4 * v + 0][2] = x[2];
                w[4 * v + 1][0] = x[3];
                w[4 * v + 1][1] = y[0];
                w[4 * v + 1][2] = y[1];
                w[4 * v + 2][0] = y[2];
                w[4 * v + 2][1] = y[3];
                w[4 * v + 2][2] = z[0];
                w[4 * v + 3][0] = z[1];
                w[4 * v + 3][1] = z[2];
                w[4 * v + 3][2] = z[3];
            }
            return PT_ALL_ITEMS;
        }
        uint32_t ok = 0;
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
            const uint32_t  i = base + j * PT_THREADS + threadIdx.x;
            const uint32_t* p = in + (size_t)min(i, end - 1u) * 3;
            w[j][0] = p[0];
            w[j][1] = p[1];
            w[j][2] = p[2];
            ok |= (uint32_t)(i < end) << j;
        }
        return ok;
    }
};

// Keys only out of 12-byte tuples (the histogram of a later pass when no digit side array was
// written): every third word.  The lines are fetched whole either way — 12 bytes of HBM traffic
// per tuple for 4 bytes of key — which pays only where the scatter behind it finds them cached.
struct Aos3KeyLoader {
    const uint32_t* in;
    __device__ __forceinline__ uint32_t key_tile(uint32_t base, uint32_t end, uint32_t (&hk)[PT_ITEMS]) const {
        if (base + PT_TILE <= end) {
#pragma unroll
            for (int v = 0; v < PT_ITEMS / 4; ++v) {
                const uint32_t* p = in + (size_t)(base + (v * PT_THREADS + threadIdx.x) * 4) * 3;
                const u32x4a    x = *reinterpret_cast<const u32x4a*>(p), y = *reinterpret_cast<const u32x4a*>(p + 4),
                             z = *reinterpret_cast<const u32x4a*>(p + 8);
                hk[4 * v + 0] = x[0];
                hk[4 * v + 1] = x[3];
                hk[4 * v + 2] = y[2];
                hk[4 * v + 3] = z[1];
            }
            return PT_ALL_ITEMS;
        }
        uint32_t ok = 0;
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
            const uint32_t i = base + j * PT_THREADS + threadIdx.x;
            hk[j] = in[(size_t)min(i, end - 1u) * 3];
            ok |= (uint32_t)(i < end) << j;
        }
        return ok;
    }
};

// The side array of 16-bit digits a 12-byte-tuple pass wrote for its successor: the histogram
// of that successor reads 2 bytes per tuple.  key_tile hands the digit back in the bit position
// the histogram kernel takes it from.
struct DigitLoader {
    static constexpr bool kNeedsBegin = true;
    const uint16_t* digits;
    uint32_t        shift;
    // Full tiles read sixteen consecutive digits per thread with two 16-byte loads.  Loads are
    // 4-byte aligned, digits 2-byte: a full tile that starts at an ODD index covers the window
    // one element earlier, [base - 1, base - 1 + PT_TILE).  The windows of consecutive tiles
    // still tile the group's range [begin, end) exactly once (the histogram does not care which
    // tile counts a tuple):
    //   * the element in front of the group's first tile (begin - 1) is masked out;
    //   * the element a full odd tile leaves behind (base + PT_TILE - 1) is the first one of the
    //     next tile's window — a partial last tile starts one element early for it, and when
    //     there is no next tile extra() hands it to thread 0.
    __device__ __forceinline__ uint32_t key_tile(uint32_t base, uint32_t begin, uint32_t end,
                                                 uint32_t (&hk)[PT_ITEMS]) const {
        static_assert(PT_ITEMS % 8 == 0, "16-byte loads of eight digits each");
        if (base + PT_TILE <= end) {
            const uint32_t* w = reinterpret_cast<const uint32_t*>(digits) + (base >> 1);
#pragma unroll
            for (int v = 0; v < PT_ITEMS / 8; ++v) {
                const uint32_t wi = (v * PT_THREADS + threadIdx.x) * 4;
                const u32x4a   x = __builtin_nontemporal_load(reinterpret_cast<const u32x4a*>(w + wi));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    hk[8 * v + 2 * e] = (x[e] & 0xffffu) << shift;
                    hk[8 * v + 2 * e + 1] = (x[e] >> 16) << shift;
                }
            }
            uint32_t ok = PT_ALL_ITEMS;
            if ((base & 1u) && base == begin && threadIdx.x == 0) ok &= ~1u;  // element begin - 1
            return ok;
        }
        const uint32_t b0 = ((base & 1u) && base != begin) ? base - 1u : base;
        uint32_t       ok = 0;
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
            const uint32_t i = b0 + j * PT_THREADS + threadIdx.x;
            hk[j] = (uint32_t)digits[min(i, end - 1u)] << shift;
            ok |= (uint32_t)(i < end) << j;
        }
        return ok;
    }
    // the group's last element when its last tile is a full odd one (see above)
    __device__ __forceinline__ bool extra(uint32_t begin, uint32_t end, uint32_t& hk) const {
        if (threadIdx.x != 0 || !(begin & 1u) || end == begin || (end - begin) % (uint32_t)PT_TILE) return false;
        hk = (uint32_t)digits[end - 1u] << shift;
        return true;
    }
};

// Partitioned tuples of ONE key word + ONE carry word can also be kept packed: an array of
// 8-byte {hashed key, carry} pairs instead of two word arrays (see k_pass_scatter_packed).
struct PackedLoader {
    const uint2* in;
    RJ_TRIVIAL_ISSUE_FINISH
    __device__ __forceinline__ uint32_t issue_keys(uint32_t base, uint32_t end, uint32_t (&lo)[PT_ITEMS],
                                                   uint32_t (&)[PT_ITEMS]) const {
        return key_tile(base, end, lo);
    }
    __device__ __forceinline__ uint32_t finish_keys(uint32_t, uint32_t, uint32_t ok, uint32_t (&)[PT_ITEMS],
                                                    uint32_t (&)[PT_ITEMS]) const {
        return ok;
    }
    // keys only (histogram of a later pass): every second word of the pairs
    __device__ __forceinline__ uint32_t key_tile(uint32_t base, uint32_t end,
                                                 uint32_t (&hk)[PT_ITEMS]) const {
        if (base + PT_TILE <= end) {
#pragma unroll
            for (int v = 0; v < PT_ITEMS / 2; ++v) {
                u32x4a x = *reinterpret_cast<const u32x4a*>(in + base + (v * PT_THREADS + threadIdx.x) * 2);
                hk[2 * v] = x[0];
                hk[2 * v + 1] = x[2];
            }
            return PT_ALL_ITEMS;
        }
        uint32_t ok = 0;
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
            uint32_t i = base + j * PT_THREADS + threadIdx.x;
            hk[j] = in[min(i, end - 1u)].x;
            ok |= (uint32_t)(i < end) << j;
        }
        return ok;
    }
    template <int NW>
    __device__ __forceinline__ uint32_t load_tile(uint32_t base, uint32_t end,
                                                  uint32_t (&w)[PT_ITEMS][NW]) const {
        static_assert(NW == 2, "packed tuples are key + one carry word");
        if (base + PT_TILE <= end) {  // two consecutive tuples per 16-byte load
#pragma unroll
            for (int v = 0; v < PT_ITEMS / 2; ++v) {
                u32x4a x = *reinterpret_cast<const u32x4a*>(in + base + (v * PT_THREADS + threadIdx.x) * 2);
                w[2 * v][0] = x[0];
                w[2 * v][1] = x[1];
                w[2 * v + 1][0] = x[2];
                w[2 * v + 1][1] = x[3];
            }
            return PT_ALL_ITEMS;
        }
        uint32_t ok = 0;
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
            uint32_t i = base + j * PT_THREADS + threadIdx.x;
            uint2    t = in[min(i, end - 1u)];
            w[j][0] = t.x;
            w[j][1] = t.y;
            ok |= (uint32_t)(i < end) << j;
        }
        return ok;
    }
};

// {hashed key, carry} pairs BETWEEN the passes of a packed plan, blocked: tuple g lives in block g / 64 —
// 128 words: the 64 keys, then the 64 carries — so that the next pass' histogram reads the key halves
// only (two of every four 128-byte lines: 4 instead of 8 bytes per tuple; with 16-tuple blocks it still
// pulled every line) while a scatter writes a run of consecutive tuples into ONE array.
struct BlockedLoader {
    const uint32_t* in;
    RJ_TRIVIAL_ISSUE_FINISH
    __device__ __forceinline__ uint32_t issue_keys(uint32_t base, uint32_t end, uint32_t (&lo)[PT_ITEMS],
                                                   uint32_t (&)[PT_ITEMS]) const {
        return key_tile(base, end, lo);
    }
    __device__ __forceinline__ uint32_t finish_keys(uint32_t, uint32_t, uint32_t ok, uint32_t (&)[PT_ITEMS],
                                                    uint32_t (&)[PT_ITEMS]) const {
        return ok;
    }
    static constexpr uint32_t BLK = 256;  // tuples per block: 1 KiB of keys, then 1 KiB of carries (64 / 256 / 1024 measured: profiles/r03_ao_*)
    static __device__ __forceinline__ size_t key_at(uint32_t g) { return (size_t)(g / BLK) * (2u * BLK) + (g % BLK); }
    __device__ __forceinline__ uint32_t key_tile(uint32_t base, uint32_t end, uint32_t (&hk)[PT_ITEMS]) const {
        uint32_t ok = 0;
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
            const uint32_t i = base + j * PT_THREADS + threadIdx.x;
            hk[j] = in[key_at(min(i, end - 1u))];
            ok |= (uint32_t)(i < end) << j;
        }
        return ok;
    }
    template <int NW>
    __device__ __forceinline__ uint32_t load_tile(uint32_t base, uint32_t end, uint32_t (&w)[PT_ITEMS][NW]) const {
        static_assert(NW == 2, "packed tuples are key + one carry word");
        uint32_t ok = 0;
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
            const uint32_t i = base + j * PT_THREADS + threadIdx.x;
            const size_t   a = key_at(min(i, end - 1u));
            w[j][0] = in[a];
            w[j][1] = in[a + BLK];
            ok |= (uint32_t)(i < end) << j;
        }
        return ok;
    }
};

// byte offset of row `row` in a 4-byte / 8-byte column (regular page images or dense)
__device__ __forceinline__ uint64_t col_off32(bool paged, uint32_t row) {
    uint32_t p = row / ROWS32, i = row - p * ROWS32;
    uint64_t po = (uint64_t)p * PAGE_BYTES + HDR32 + i * 4u;
    return paged ? po : (uint64_t)row * 4u;
}
__device__ __forceinline__ uint64_t col_off64(bool paged, uint32_t row) {
    uint32_t p = row / ROWS64, i = row - p * ROWS64;
    uint64_t po = (uint64_t)p * PAGE_BYTES + HDR64 + i * 8u;
    return paged ? po : (uint64_t)row * 8u;
}

// Forms tuples straight from columns: page decode (regular pages), NULL-key
// drop (reference src/execute.cpp:62-83: only rows whose variant holds KeyType
// are valid) and hashing fused into the first radix pass.
// WIDE (WideLayout): the carry is made of several columns — two or three 32-bit ones, or a
// 64-bit one followed by a 32-bit one (CARRY_WIDE; a validity word, where some carried column
// has NULLs, is one of the 32-bit columns by then: k_pack_validity).
template <int KW, int CW, int WIDE = WIDE_NONE>
struct SrcLoader {
    static_assert(WIDE == WIDE_NONE || (WIDE == WIDE_32S && CW >= 2) || (WIDE == WIDE_64_32 && CW == 3), "wide carry layouts");
    TupleSrc s;

    // Row of item j.  Full tiles use the vector mapping (four consecutive rows per thread
    // and 16-byte load; ROWS32 % 4 == 0 and tiles start at multiples of 4, so the four rows
    // share a page), partial tiles the strided one with clamped indices.
    __device__ __forceinline__ static uint32_t item_row(bool vec, uint32_t base, int j) {
        return vec ? base + ((j / 4) * PT_THREADS + threadIdx.x) * 4 + (j % 4)
                   : base + j * PT_THREADS + threadIdx.x;
    }

    // 32-bit column -> one word per item
    // NT: non-temporal loads — the histogram kernels stream the key column once and gain from
    // not parking it in the L2 (pass-1 histogram 1.47 -> 1.38 ms at 1 B rows, the fine histogram
    // 0.115 -> 0.085 ms at 100 M); the scatter kernels lose with the same hint (their loads
    // compete with half-written output lines for the L2: profiles/r02_ax_*), so they do not use it
    // Addresses are formed as a UNIFORM 64-bit tile base (the page that holds the tile's first row,
    // or the first row itself for dense columns) plus a 32-bit per-lane byte offset — one VGPR per
    // outstanding load instead of a 64-bit pair, and no 64-bit vector arithmetic: with all loads of
    // a tile in flight at once the 64-bit form spilled to scratch.
    template <bool NT = false>
    __device__ __forceinline__ static void load_col32(const ColRef& c, bool vec, uint32_t base,
                                                      uint32_t end, uint32_t (&out)[PT_ITEMS]) {
        const bool     paged = c.kind == COL_PAGED;
        const uint32_t ub = __builtin_amdgcn_readfirstlane(base);
        const uint32_t p0 = ub / ROWS32, s0 = ub - p0 * ROWS32;  // page / slot of the tile's first row
        const uint8_t* tp = paged ? c.ptr + (size_t)p0 * PAGE_BYTES : c.ptr + (size_t)ub * 4u;
        // byte offset of row base + rel from tp, branch-free for both kinds:
        //   dense  rel * 4
        //   paged  q * PAGE + HDR + (t - q * ROWS) * 4  =  rel * 4 + (s0 * 4 + HDR) + q * (PAGE - ROWS * 4),
        //          t = s0 + rel, q = t / ROWS  (pages crossed since the tile's first page)
        const uint32_t bias = paged ? s0 * 4u + HDR32 : 0u, per_page = paged ? PAGE_BYTES - ROWS32 * 4u : 0u;
        auto           off = [&](uint32_t rel) -> uint32_t { return rel * 4u + bias + ((s0 + rel) / ROWS32) * per_page; };
        if (vec) {
#pragma unroll
            for (int v = 0; v < PT_ITEMS / 4; ++v) {
                const u32x4a* q = reinterpret_cast<const u32x4a*>(tp + off((v * PT_THREADS + threadIdx.x) * 4u));
                u32x4a        x;
                if constexpr (NT)
                    x = __builtin_nontemporal_load(q);
                else
                    x = *q;
#pragma unroll
                for (int e = 0; e < 4; ++e) out[4 * v + e] = x[e];
            }
        } else {
            const uint32_t last = end - 1u - ub;
#pragma unroll
            for (int j = 0; j < PT_ITEMS; ++j)
                out[j] = *reinterpret_cast<const uint32_t*>(tp + off(min((uint32_t)(j * PT_THREADS + threadIdx.x), last)));
        }
    }
    template <bool NT = false>
    __device__ __forceinline__ static void load_col64(const ColRef& c, bool vec, uint32_t base,
                                                      uint32_t end, uint32_t (&lo)[PT_ITEMS],
                                                      uint32_t (&hi)[PT_ITEMS]) {
        const bool     paged = c.kind == COL_PAGED;
        const uint32_t ub = __builtin_amdgcn_readfirstlane(base);
        const uint32_t p0 = ub / ROWS64, s0 = ub - p0 * ROWS64;
        const uint8_t* tp = paged ? c.ptr + (size_t)p0 * PAGE_BYTES : c.ptr + (size_t)ub * 8u;
        const uint32_t last = end - 1u - ub;
        const uint32_t bias = paged ? s0 * 8u + HDR64 : 0u, per_page = paged ? PAGE_BYTES - ROWS64 * 8u : 0u;
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
            const uint32_t rel = min(item_row(vec, 0u, j), last);
            const uint32_t o = rel * 8u + bias + ((s0 + rel) / ROWS64) * per_page;  // (see load_col32)
            const uint64_t* a = reinterpret_cast<const uint64_t*>(tp + o);
            uint64_t        v;
            if constexpr (NT)
                v = __builtin_nontemporal_load(a);
            else
                v = *a;
            lo[j] = (uint32_t)v;
            hi[j] = (uint32_t)(v >> 32);
        }
    }
    // a 32-bit column of any kind (a base table's row-id column included)
    __device__ __forceinline__ static void load_col32_any(const ColRef& c, bool vec, uint32_t base, uint32_t end,
                                                          uint32_t (&out)[PT_ITEMS]) {
        if (c.kind == COL_IOTA) {
#pragma unroll
            for (int j = 0; j < PT_ITEMS; ++j) out[j] = item_row(vec, base, j);
        } else {
            load_col32(c, vec, base, end, out);
        }
    }

    // raw key words of the tile + in-range mask
    template <bool NT = false>
    __device__ __forceinline__ uint32_t raw_keys(bool vec, uint32_t base, uint32_t end,
                                                 uint32_t (&lo)[PT_ITEMS],
                                                 uint32_t (&hi)[PT_ITEMS]) const {
        if constexpr (KW == 1) {
            load_col32<NT>(s.key, vec, base, end, lo);
#pragma unroll
            for (int j = 0; j < PT_ITEMS; ++j) hi[j] = 0;
        } else {
            load_col64<NT>(s.key, vec, base, end, lo, hi);
        }
        if (vec) return PT_ALL_ITEMS;
        uint32_t ok = 0;
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j)
            ok |= (uint32_t)(base + j * PT_THREADS + threadIdx.x < end) << j;
        return ok;
    }
    __device__ __forceinline__ uint32_t drop_invalid(bool vec, uint32_t base, uint32_t end,
                                                     uint32_t ok) const {
        const uint8_t* vp = s.key.valid;
        if (vp) {  // uniform: the column was decoded by K1 and carries validity bytes
#pragma unroll
            for (int j = 0; j < PT_ITEMS; ++j)
                if (!vp[min(item_row(vec, base, j), end - 1u)]) ok &= ~(1u << j);
        }
        return ok;
    }
    // hash in place; FP64: NaN never matches (see key semantics below)
    __device__ __forceinline__ uint32_t hash_keys(uint32_t ok, uint32_t (&lo)[PT_ITEMS],
                                                  uint32_t (&hi)[PT_ITEMS]) const {
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
            if constexpr (KW == 1) {
                lo[j] = s.prehashed ? lo[j] : fmix32(lo[j]);
            } else {
                uint64_t k = (uint64_t)lo[j] | ((uint64_t)hi[j] << 32);
                // The reference hashes the BIT PATTERN of a double (src/execute.cpp:28-31) and
                // compares with == (:215,231): NaN equals nothing, and -0.0 / +0.0 hash to
                // different slots so they only meet by accident of probing.  Bit-pattern
                // equality with NaN excluded reproduces that (SURVEY.md §8a note on FP64).
                if (s.key_f64 && (k & 0x7ff0000000000000ull) == 0x7ff0000000000000ull &&
                    (k & 0x000fffffffffffffull))
                    ok &= ~(1u << j);
                uint64_t h = fmix64(k);
                lo[j] = (uint32_t)h;
                hi[j] = (uint32_t)(h >> 32);
            }
        }
        return ok;
    }
    __device__ __forceinline__ uint32_t key_tile(uint32_t base, uint32_t end,
                                                 uint32_t (&hk)[PT_ITEMS]) const {
        const bool vec = base + PT_TILE <= end;
        uint32_t   hi[PT_ITEMS];
        uint32_t   ok = raw_keys<true>(vec, base, end, hk, hi);
        ok = drop_invalid(vec, base, end, ok);
        return hash_keys(ok, hk, hi);
    }
    // The same in two halves, for a software pipeline: issue_keys only ISSUES the loads of the
    // raw key words, finish_keys (which needs the data) drops invalid rows and hashes.
    __device__ __forceinline__ uint32_t issue_keys(uint32_t base, uint32_t end,
                                                   uint32_t (&lo)[PT_ITEMS],
                                                   uint32_t (&hi)[PT_ITEMS]) const {
        return raw_keys<true>(base + PT_TILE <= end, base, end, lo, hi);
    }
    __device__ __forceinline__ uint32_t finish_keys(uint32_t base, uint32_t end, uint32_t ok,
                                                    uint32_t (&lo)[PT_ITEMS],
                                                    uint32_t (&hi)[PT_ITEMS]) const {
        ok = drop_invalid(base + PT_TILE <= end, base, end, ok);
        return hash_keys(ok, lo, hi);
    }
    // The tile in two halves (see issue_keys / finish_keys): issue_tile ISSUES every load of the
    // tile — raw key words into w[.][0..KW), carry words behind them — and finish_tile, which
    // needs the key data, drops invalid rows and hashes the keys in place.
    template <int NW>
    __device__ __forceinline__ uint32_t issue_tile(uint32_t base, uint32_t end,
                                                   uint32_t (&w)[PT_ITEMS][NW]) const {
        static_assert(NW == KW + CW, "word count");
        const bool vec = base + PT_TILE <= end;
        uint32_t   lo[PT_ITEMS], hi[PT_ITEMS];
        uint32_t   ok = raw_keys(vec, base, end, lo, hi);
        if constexpr (WIDE == WIDE_32S) {
            uint32_t c0[PT_ITEMS];
            load_col32_any(s.carry, vec, base, end, c0);
#pragma unroll
            for (int j = 0; j < PT_ITEMS; ++j) w[j][KW] = c0[j];
            load_col32_any(s.carry2, vec, base, end, c0);
#pragma unroll
            for (int j = 0; j < PT_ITEMS; ++j) w[j][KW + 1] = c0[j];
            if constexpr (CW == 3) {
                load_col32_any(s.carry3, vec, base, end, c0);
#pragma unroll
                for (int j = 0; j < PT_ITEMS; ++j) w[j][KW + 2] = c0[j];
            }
        } else if constexpr (WIDE == WIDE_64_32) {
            uint32_t c0[PT_ITEMS], c1[PT_ITEMS];
            load_col64(s.carry, vec, base, end, c0, c1);
#pragma unroll
            for (int j = 0; j < PT_ITEMS; ++j) {
                w[j][KW] = c0[j];
                w[j][KW + 1] = c1[j];
            }
            load_col32_any(s.carry2, vec, base, end, c0);
#pragma unroll
            for (int j = 0; j < PT_ITEMS; ++j) w[j][KW + 2] = c0[j];
        } else if constexpr (CW >= 1) {
            if (s.carry_mode == CARRY_ROWIDX) {  // CW == 1 by construction
#pragma unroll
                for (int j = 0; j < PT_ITEMS; ++j) w[j][KW] = item_row(vec, base, j);
            } else if constexpr (CW == 1) {
                uint32_t c0[PT_ITEMS];
                load_col32(s.carry, vec, base, end, c0);
#pragma unroll
                for (int j = 0; j < PT_ITEMS; ++j) w[j][KW] = c0[j];
            } else {
                uint32_t c0[PT_ITEMS], c1[PT_ITEMS];
                load_col64(s.carry, vec, base, end, c0, c1);
#pragma unroll
                for (int j = 0; j < PT_ITEMS; ++j) {
                    w[j][KW] = c0[j];
                    w[j][KW + 1] = c1[j];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
            w[j][0] = lo[j];
            if constexpr (KW == 2) w[j][1] = hi[j];
        }
        return ok;
    }
    template <int NW>
    __device__ __forceinline__ uint32_t finish_tile(uint32_t base, uint32_t end, uint32_t ok,
                                                    uint32_t (&w)[PT_ITEMS][NW]) const {
        const bool vec = base + PT_TILE <= end;
        uint32_t   lo[PT_ITEMS], hi[PT_ITEMS];
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
            lo[j] = w[j][0];
            hi[j] = KW == 2 ? w[j][KW - 1] : 0u;
        }
        ok = drop_invalid(vec, base, end, ok);
        ok = hash_keys(ok, lo, hi);
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
            w[j][0] = lo[j];
            if constexpr (KW == 2) w[j][1] = hi[j];
        }
        return ok;
    }
    template <int NW>
    __device__ __forceinline__ uint32_t load_tile(uint32_t base, uint32_t end,
                                                  uint32_t (&w)[PT_ITEMS][NW]) const {
        return finish_tile<NW>(base, end, issue_tile<NW>(base, end, w), w);
    }
};

// Which tuples does workgroup `g` own?  A group is up to tiles_per_group
// consecutive tiles inside ONE input segment.
__device__ __forceinline__ bool group_range(const PassParams& pp, uint32_t g, uint32_t& seg,
                                            uint32_t& begin, uint32_t& end) {
    const uint32_t gt = pp.tiles_per_group * (uint32_t)PT_TILE;
    if (pp.seg_off == nullptr) {
        seg = 0;
        uint64_t b = (uint64_t)g * gt;
        if (b >= pp.n) return false;
        begin = (uint32_t)b;
        end = (uint32_t)min((uint64_t)pp.n, b + gt);
        return true;
    }
    const uint32_t G = pp.grp_start[pp.nseg] - pp.grp_base;
    if (pp.xcd_remap) {
        const uint32_t per = (G + 7u) >> 3;
        if ((g >> 3) >= per) return false;
        g = (g & 7u) * per + (g >> 3);
    }
    if (g >= G) return false;
    uint32_t lo = 0, hi = pp.nseg;  // largest s with grp_start[s] <= g
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (pp.grp_start[mid] - pp.grp_base <= g)
            lo = mid;
        else
            hi = mid;
    }
    seg = lo >> pp.oseg_shift;  // the OUTPUT segment (bins, cursors) input segment `lo` feeds
    uint32_t gi = g - (pp.grp_start[lo] - pp.grp_base);
    uint64_t b = (uint64_t)pp.seg_off[lo] + (uint64_t)gi * gt;
    begin = (uint32_t)b;
    end = (uint32_t)min((uint64_t)(pp.seg_end ? pp.seg_end[lo] : pp.seg_off[lo + 1]), b + gt);
    return begin < end;
}

template <class L, class = void>
struct loader_needs_begin : std::false_type {};
template <class L>
struct loader_needs_begin<L, std::enable_if_t<L::kNeedsBegin>> : std::true_type {};

// ============================================================== K2 histogram
// Counting half of the radix partition (reference counterpart: the serial
// histogram src/execute.cpp:124-132).  LDS atomics per tuple, F global adds per group for
// the bin totals.
template <class Loader>
__global__ __launch_bounds__(PT_THREADS) void k_pass_hist(Loader ld, PassParams pp) {
    __shared__ uint32_t s_h[PT_MAXF];
#if RJ_PT_HOT_PROBE
    __shared__ uint32_t s_hot[4096];
    for (uint32_t d = threadIdx.x; d < 4096; d += PT_THREADS) s_hot[d] = 0xffffffffu - d;
#endif
    uint32_t            seg, begin, end;
    if (!group_range(pp, blockIdx.x, seg, begin, end)) return;
    const uint32_t F = 1u << pp.fanout_log2, mask = F - 1u;
    for (uint32_t d = threadIdx.x; d < F; d += PT_THREADS) s_h[d] = 0;
    lds_barrier();
    for (uint32_t base = begin; base < end; base += PT_TILE) {
        uint32_t hk[PT_ITEMS];
        uint32_t ok;
        if constexpr (loader_needs_begin<Loader>::value)
            ok = ld.key_tile(base, begin, end, hk);
        else
            ok = ld.key_tile(base, end, hk);
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
#if RJ_PT_HOT_PROBE
            if (s_hot[(hk[j] >> 9) & 4095u] == hk[j]) ok &= ~(1u << j);
#endif
            if ((ok >> j) & 1u) atomicAdd(&s_h[pass_digit(pp, hk[j], mask)], 1u);
        }
    }
    if constexpr (loader_needs_begin<Loader>::value) {
        uint32_t hx;
        if (ld.extra(begin, end, hx)) atomicAdd(&s_h[pass_digit(pp, hx, mask)], 1u);
    }
    lds_barrier();
    for (uint32_t d = threadIdx.x; d < F; d += PT_THREADS) {
        uint32_t c = s_h[d];
        if (c) atomicAdd(&pp.hist[(((size_t)seg * F + d) << pp.xcd_log2) | (blockIdx.x & ((1u << pp.xcd_log2) - 1u))], c);
    }
}

// Two-pass plans whose final partition count fits an LDS histogram (<= 2^PT_FINEBITS bins)
// count BOTH digits in the one read of the source: bin q = d1 * F2 + d2 is the final
// partition, its exclusive scan is at once the pass-2 offsets and (every F2-th entry) the
// pass-1 offsets, and no second histogram pass over the partitioned tuples is needed.
// Persistent workgroups (one per CU: the histogram takes most of the LDS) stride over the
// tiles and flush their bins with one global add each.
// xcd_tpg > 0 (XCD-aware placement, see PassParams::xcd_log2): the first pass' scatter works in
// groups of xcd_tpg tiles and writes group g's runs into sub-range g & 7 of every partition, so
// it needs the pass-1 digit counts per sub-range.  Workgroup w therefore counts exactly the tile
// groups g with g & 7 == w & 7 (gridDim.x is a multiple of 8) — its whole LDS histogram belongs
// to ONE sub-range, and the per-sub-range counts fall out of the flush (row sums of its bins)
// with no extra work per tuple.
template <class Loader>
__global__ __launch_bounds__(PT_THREADS) void k_fine_hist(Loader ld, uint32_t n, uint32_t shift,
                                                          uint32_t b1, uint32_t b2,
                                                          uint32_t* fine, uint32_t xcd_tpg,
                                                          uint32_t* coarse_x) {
    __shared__ uint32_t s_f[1u << PT_FINEBITS];
    const uint32_t      NB = 1u << (b1 + b2), m1 = (1u << b1) - 1u, m2 = (1u << b2) - 1u;
    for (uint32_t d = threadIdx.x; d < NB; d += PT_THREADS) s_f[d] = 0;
    lds_barrier();
    const uint32_t tiles = (uint32_t)(((uint64_t)n + PT_TILE - 1) / PT_TILE);
    // tile walk: plain striding, or the tile groups of this workgroup's sub-range
    const uint32_t x = blockIdx.x & 7u, gstride = gridDim.x >> 3;
    uint32_t       gk = blockIdx.x >> 3, gj = 0;  // group x + 8 * gk, tile gj inside it
    auto           tile_of = [&](uint32_t k, uint32_t j) { return (x + 8u * k) * xcd_tpg + j; };
    auto           advance = [&](uint32_t t) -> uint32_t {  // the tile after t, >= tiles: none
        if (!xcd_tpg) return t + gridDim.x;
        if (++gj < xcd_tpg && tile_of(gk, gj) < tiles) return tile_of(gk, gj);
        gk += gstride;
        gj = 0;
        const uint64_t nt = (uint64_t)(x + 8ull * gk) * xcd_tpg;
        return nt < tiles ? (uint32_t)nt : 0xffffffffu;
    };
    uint32_t t = xcd_tpg ? tile_of(gk, 0) : blockIdx.x;
    // Software pipeline: the LDS atomics of a tile (a third of its time) run while the NEXT
    // tile's loads are in flight — the raw loads are issued before the atomics, hashing (which
    // needs the data) comes after them.
    uint32_t lo[PT_ITEMS], hi[PT_ITEMS], raw = 0;
    if (t < tiles) raw = ld.issue_keys(t * (uint32_t)PT_TILE, n, lo, hi);
    while (t < tiles) {
        const uint32_t ok = ld.finish_keys(t * (uint32_t)PT_TILE, n, raw, lo, hi);
        uint32_t       q[PT_ITEMS];
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
            const uint32_t h = lo[j] >> shift;
            q[j] = ((h & m1) << b2) | ((h >> b1) & m2);
        }
        const uint32_t t2 = advance(t);
        if (t2 < tiles) raw = ld.issue_keys(t2 * (uint32_t)PT_TILE, n, lo, hi);
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j)
            if ((ok >> j) & 1u) atomicAdd(&s_f[q[j]], 1u);
        t = t2;
    }
    lds_barrier();
    for (uint32_t d = threadIdx.x; d < NB; d += PT_THREADS) {
        uint32_t c = s_f[d];
        if (c) atomicAdd(&fine[d], c);
    }
    if (xcd_tpg) {
        const uint32_t F2 = 1u << b2;
        for (uint32_t d1 = threadIdx.x; d1 <= m1; d1 += PT_THREADS) {
            uint32_t sum = 0;
            // (rotated start: the threads of a wave do not all hit the same LDS bank)
            for (uint32_t i = 0; i < F2; ++i) sum += s_f[d1 * F2 + ((i + d1) & m2)];
            if (sum) atomicAdd(&coarse_x[(d1 << 3) | x], sum);
        }
    }
}

// Exclusive scan of the fine histogram, one workgroup per pass-1 digit: the workgroup sums
// everything below its F2 bins (coalesced, the whole histogram is 128 KiB of L2), scans its
// own bins on top, and with that writes the pass-2 offsets/cursors of its segment and the
// pass-1 offset/cursor of its digit — one launch instead of a serial walk over 2^15 bins.
__global__ __launch_bounds__(PT_MAXF) void k_scan_fine(const uint32_t* fine, uint32_t F1,
                                                       uint32_t F2, uint32_t* off2,
                                                       uint32_t* cursor2, uint32_t* off1,
                                                       uint32_t* cursor1, const uint32_t* coarse_x) {
    __shared__ uint32_t s_wsum[PT_MAXF / 64];
    const uint32_t      d1 = blockIdx.x, below = d1 * F2;
    uint32_t            sum = 0;
    if ((F2 & 3u) == 0) {  // 16 bytes per load (hipMalloc'd, so `fine` is 16-byte aligned)
        const uint4* f4 = reinterpret_cast<const uint4*>(fine);
        for (uint32_t i = threadIdx.x; i < below / 4; i += PT_MAXF) {
            const uint4 v = f4[i];
            sum += v.x + v.y + v.z + v.w;
        }
    } else {
        for (uint32_t i = threadIdx.x; i < below; i += PT_MAXF) sum += fine[i];
    }
    uint32_t base, tot;
    (void)block_excl_scan(sum, s_wsum, base);
    lds_barrier();
    const uint32_t v = threadIdx.x < F2 ? fine[below + threadIdx.x] : 0u;
    const uint32_t ex = block_excl_scan(v, s_wsum, tot);
    if (threadIdx.x < F2) {
        off2[below + threadIdx.x] = base + ex;
        cursor2[below + threadIdx.x] = base + ex;
    }
    if (threadIdx.x == 0) {
        off1[d1] = base;
        if (coarse_x) {  // eight sub-ranges (one per XCD) one behind the other: cursor1[d1 * 8 + x]
            uint32_t at = base;
            for (uint32_t x = 0; x < 8; ++x) {
                cursor1[(d1 << 3) | x] = at;
                at += coarse_x[(d1 << 3) | x];
            }
        } else {
            cursor1[d1] = base;
        }
        if (d1 + 1 == F1) {
            off1[F1] = base + tot;
            off2[(size_t)F1 * F2] = base + tot;
        }
    }
}

// ================================================================ K4 scatter
// Scatter half of the radix partition (reference counterpart: the serial
// scatter src/execute.cpp:175-184).  Per tile of PT_TILE (16384) tuples:
//   1. every tuple takes its rank inside its digit from an LDS atomic add;
//   2. the digit counters are scanned (wave shuffles) into LDS positions;
//   3. per word array the tile is written to LDS in digit order, then copied out
//      so that consecutive lanes write consecutive addresses of one digit's run
//      (software write-combining: HBM sees contiguous runs, not 4-byte scatters).
// Every tile reserves its output ranges with one atomic per digit (cursor = start of the
// digit's partition, from the scanned histogram).
// PAIR >= 0: words PAIR and PAIR+1 (a two-word carry) are written as 8-byte pairs into ONE array,
// out.w[PAIR] — one output stream and one staging round less than two word arrays.
// AOS (NW == 3, PAIR == 1: key + two-word carry, the LAST pass of a plan): the output is ONE array
// of 12-byte {hashed key, carry lo, carry hi} tuples instead of a key array plus a pair array —
// one output stream whose runs are 12 bytes per tuple long (a 32-tuple run is 384 contiguous
// bytes instead of 128 + 256 in two places), and the join reads one stream.  The tile is staged
// in two halves by sorted position (8192 tuples = 96 KiB of LDS each), so the runs are still
// those of the whole 16384-tuple tile.
#ifndef RJ_PT_DIAG
#define RJ_PT_DIAG 0
#endif
// Tile prefetch in the key + two-word-carry scatter (next tile's loads issued behind the second
// half's staging): measured SLOWER at 1 B rows — pass 1 10.4-11.3 -> 12.0 ms, pass 2 9.5 -> 10.2 ms per
// step (profiles/r03_h_scatter_tile_prefetch_ab.log): the kernels are bound by the memory system's
// mix of reads and partial-line writes, not by a CU's load latency, and a burst of loads next to
// the copy-out's stores only gets in their way.  Off; -DRJ_PT_PIPELINE=1 builds it for A/B.
#ifndef RJ_PT_EXTRA_LOOKUP
#define RJ_PT_EXTRA_LOOKUP 0
#endif
// (experiment: the per-tuple cost of a HOT-KEY test — one probe of a 4096-entry LDS set of hashed keys —
// in the first pass' histogram and scatter, which a skew bypass (hot probe keys joined straight out of the
// first pass instead of being partitioned twice) would pay for every tuple.  The set is empty here.)
#ifndef RJ_PT_HOT_PROBE
#define RJ_PT_HOT_PROBE 0
#endif
#ifndef RJ_PT_PIPELINE
#define RJ_PT_PIPELINE 0
#endif
#if RJ_PT_DIAG
#define RJ_PT_STAMP(PHASE)                                                      \
    do {                                                                        \
        if (pp.diag) {                                                          \
            unsigned long long _t = __builtin_amdgcn_s_memtime();               \
            if (threadIdx.x == 0) atomicAdd(&pp.diag[PHASE], _t - diag_t);      \
            diag_t = _t;                                                        \
        }                                                                       \
    } while (0)
#else
#define RJ_PT_STAMP(PHASE) \
    do {                   \
    } while (0)
#endif

template <int NW, class Loader, int PAIR, bool AOS>
__global__ __launch_bounds__(PT_THREADS) void k_pass_scatter(Loader ld, PassParams pp, Words out) {
    static_assert(PAIR < 0 || (PAIR >= 1 && PAIR + 1 < NW), "pair = two carry words behind the key");
    static_assert(!AOS || (NW == 3 && PAIR == 1), "12-byte tuples: one key word + a two-word carry");
    __shared__ uint2    s_stage2[PT_TILE];  // 128 KiB: a word array uses the first half
    uint32_t* const     s_stage = reinterpret_cast<uint32_t*>(s_stage2);
    __shared__ uint32_t s_cnt[PT_MAXF];
    __shared__ uint32_t s_base[PT_MAXF];
    __shared__ uint32_t s_delta[PT_MAXF];
    __shared__ uint32_t s_wsum[PT_THREADS / 64];
#if RJ_PT_HOT_PROBE
    __shared__ uint32_t s_hot[4096];
    for (uint32_t d = threadIdx.x; d < 4096; d += PT_THREADS) s_hot[d] = 0xffffffffu - d;
#endif
    uint32_t            seg, begin, end;
    if (!group_range(pp, blockIdx.x, seg, begin, end)) return;
    const uint32_t F = 1u << pp.fanout_log2, mask = F - 1u;

    // Thread d owns digit d's write cursor in a REGISTER: the reservation's round trip is
    // not waited for until the first tile has been loaded and ranked.
    uint32_t run = 0;  // thread d: where digit d's run of the current tile starts in the output
#if RJ_PT_EXTRA_LOOKUP
    uint32_t lookup = 0;
#define RJ_PT_USE_LOOKUP()                  \
    do {                                    \
        asm volatile("" : "+v"(lookup));    \
        run += lookup & 0u;                 \
    } while (0)
#else
#define RJ_PT_USE_LOOKUP() \
    do {                   \
    } while (0)
#endif
#if RJ_PT_DIAG
    unsigned long long diag_t = pp.diag ? __builtin_amdgcn_s_memtime() : 0ull;
#endif

    // (RJ_PT_PIPELINE, off — see above: the tile's registers are dead once its second half has been
    // staged, so the NEXT tile's loads could be issued there and fly during that half's copy-out)
    constexpr bool PIPE = NW == 3 && PAIR == 1 && RJ_PT_PIPELINE;
    uint32_t       w[PT_ITEMS][NW];
    uint32_t       ok_raw = 0;
    if constexpr (PIPE) ok_raw = ld.template issue_tile<NW>(begin, end, w);

    for (uint32_t base = begin; base < end; base += PT_TILE) {
        for (uint32_t d = threadIdx.x; d < F; d += PT_THREADS) s_cnt[d] = 0;
        lds_barrier();
        RJ_PT_STAMP(0);  // counters cleared (+ the previous tile's tail)

        uint32_t dr[PT_ITEMS];  // digit << 16 | rank, 0xffffffff = no tuple
        // every load of the tile is issued before the first rank is taken
        uint32_t ok;
        if constexpr (PIPE)
            ok = ld.template finish_tile<NW>(base, end, ok_raw, w);
        else
            ok = ld.template load_tile<NW>(base, end, w);
        RJ_PT_STAMP(1);  // loads issued (+ hashing, which waits for the key loads)
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
            dr[j] = 0xffffffffu;
#if RJ_PT_HOT_PROBE
            if (s_hot[(w[j][0] >> 9) & 4095u] == w[j][0]) ok &= ~(1u << j);
#endif
            if ((ok >> j) & 1u) {
                uint32_t d = pass_digit(pp, w[j][0], mask);
                uint32_t r = atomicAdd(&s_cnt[d], 1u);
                dr[j] = (d << 16) | r;
            }
        }
        lds_barrier();
        RJ_PT_STAMP(2);  // ranked (LDS atomics) + barrier

        // PT_THREADS >= PT_MAXF: thread d scans digit d
        // Thread d reserves digit d's range of this tile with one global atomic; its round
        // trip is not waited for until word 0 has been staged.
        uint32_t c = threadIdx.x < F ? s_cnt[threadIdx.x] : 0u;
        if (c) {
            run = atomicAdd(&pp.cursor[(((size_t)seg * F + threadIdx.x) << pp.xcd_log2) |
                                       (blockIdx.x & ((1u << pp.xcd_log2) - 1u))], c);
#if RJ_PT_EXTRA_LOOKUP
            // (experiment: what a SECOND dependent global round trip per digit and tile would cost — the
            // chunk-table lookup of a first pass that reserves from chunk lists instead of a histogram's
            // exact ranges; reads a word of the (by now read-only) bin totals at an index that depends on the
            // reservation and folds nothing into it.  Reading the cursor array itself, which every
            // workgroup's atomics keep hot, tripled the first scatter: a chunk table must not share lines
            // with the cursors)
            lookup = __builtin_nontemporal_load(&pp.hist[((size_t)(run >> 16) % ((size_t)F << pp.xcd_log2))]);
#endif
        }
        uint32_t total;
        uint32_t ex = block_excl_scan(c, s_wsum, total);
        if (threadIdx.x < F) s_base[threadIdx.x] = ex;
        lds_barrier();
        RJ_PT_STAMP(3);  // reservation issued + digit scan

        // LDS position of every tuple, computed once for all word arrays
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j)
            if (dr[j] != 0xffffffffu) dr[j] = s_base[dr[j] >> 16] + (dr[j] & 0xffffu);

        // key + two-word carry (NW == 3): keys and carry pairs of half the sorted tile are staged
        // TOGETHER and copied out together — as 12-byte tuples (AOS) or into the key array and the
        // pair array — so no per-element destination has to be kept in registers between the two
        // arrays (the 16 extra VGPRs spilled to scratch)
        if constexpr (NW == 3 && PAIR == 1) {
            constexpr uint32_t HALF = PT_TILE / 2;
            uint2* const       s_p = s_stage2;                                      // [HALF] carries
            uint32_t* const    s_k = reinterpret_cast<uint32_t*>(s_stage2 + HALF);  // [HALF] keys
            RJ_PT_USE_LOOKUP();
            if (threadIdx.x < F) s_delta[threadIdx.x] = run - ex;  // global index = delta + sorted position
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int j = 0; j < PT_ITEMS; ++j) {
                    const uint32_t lp = dr[j] - h * HALF;  // (no tuple: 0xffffffff stays out of range)
                    if (lp < HALF) {
                        s_k[lp] = w[j][0];
                        s_p[lp] = make_uint2(w[j][1], w[j][2]);
                    }
                }
                lds_barrier();
                RJ_PT_STAMP(4);  // (AOS) half staged (waits for the carry loads)
                if constexpr (PIPE) {
                    if (h == 1 && base + PT_TILE < end) ok_raw = ld.template issue_tile<NW>(base + PT_TILE, end, w);
                }
#pragma unroll
                for (int k = 0; k < PT_ITEMS / 2; ++k) {
                    const uint32_t i = k * PT_THREADS + threadIdx.x, gi = h * HALF + i;
                    if (gi < total) {
                        const uint32_t v = s_k[i];
                        const uint2    c = s_p[i];
                        const uint32_t g = s_delta[pass_digit(pp, v, mask)] + gi;
                        if constexpr (AOS) {
                            uint32_t* o = out.w[0] + (size_t)g * 3u;
                            o[0] = v;
                            o[1] = c.x;
                            o[2] = c.y;
                            if (pp.side_out) pp.side_out[g] = (uint16_t)((v >> pp.next_shift) & pp.next_mask);
                        } else {
                            out.w[0][g] = v;
                            reinterpret_cast<uint2*>(out.w[1])[g] = c;
                        }
                    }
                }
                lds_barrier();
                RJ_PT_STAMP(5);  // (AOS) half copied out (stores issued)
            }
            continue;
        }
        // word 0 (the hashed key): stage in digit order, copy out.  The digit of a staged
        // key is recomputed from the key itself, and the global destination of each LDS
        // position is kept in a register for the remaining word arrays.
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j)
            if (dr[j] != 0xffffffffu) s_stage[dr[j]] = w[j][0];
        RJ_PT_USE_LOOKUP();
        if (threadIdx.x < F) s_delta[threadIdx.x] = run - ex;  // global index = delta + LDS position
        lds_barrier();
        RJ_PT_STAMP(4);  // word 0 staged
        uint32_t dest[PT_ITEMS];
#pragma unroll
        for (int k = 0; k < PT_ITEMS; ++k) {
            const uint32_t i = k * PT_THREADS + threadIdx.x;
            dest[k] = 0;
            if (i < total) {
                const uint32_t v = s_stage[i];
                dest[k] = s_delta[pass_digit(pp, v, mask)] + i;
                out.w[0][dest[k]] = v;
            }
        }
        lds_barrier();
        RJ_PT_STAMP(5);  // word 0 copied out (stores issued)
#pragma unroll
        for (int a = 1; a < NW; ++a) {
            if constexpr (PAIR >= 0) {
                if (a == PAIR + 1) continue;  // went out with word PAIR
                if (a == PAIR) {
#pragma unroll
                    for (int j = 0; j < PT_ITEMS; ++j)
                        if (dr[j] != 0xffffffffu)
                            s_stage2[dr[j]] = make_uint2(w[j][a], w[j][a + 1 < NW ? a + 1 : a]);
                    lds_barrier();
                    RJ_PT_STAMP(6);  // carry pair staged (waits for the carry loads)
                    uint2* dst2 = reinterpret_cast<uint2*>(out.w[a]);
#pragma unroll
                    for (int k = 0; k < PT_ITEMS; ++k) {
                        const uint32_t i = k * PT_THREADS + threadIdx.x;
                        if (i < total) dst2[dest[k]] = s_stage2[i];
                    }
                    lds_barrier();
                    RJ_PT_STAMP(7);  // carry pair copied out (stores issued)
                    continue;
                }
            }
#pragma unroll
            for (int j = 0; j < PT_ITEMS; ++j)
                if (dr[j] != 0xffffffffu) s_stage[dr[j]] = w[j][a];
            lds_barrier();
            uint32_t* dst = out.w[a];
#pragma unroll
            for (int k = 0; k < PT_ITEMS; ++k) {
                const uint32_t i = k * PT_THREADS + threadIdx.x;
                if (i < total) dst[dest[k]] = s_stage[i];
            }
            lds_barrier();
        }
    }
}

// Packed variant for {hashed key, one carry word} tuples (the BASELINE shape): the tile is
// staged as 8-byte pairs (128 KiB of LDS — the workgroup owns the CU anyway) and copied out
// with one 8-byte store per tuple into ONE output array.  Against two word arrays this
// halves the number of output streams and doubles the bytes per contiguous run (a run of 64
// tuples is 512 contiguous bytes instead of two runs of 256), which is what the scatter's
// rate follows; it also needs one staging round and two barriers less per tile.
template <class Loader, bool BLK_OUT = false>
__global__ __launch_bounds__(PT_THREADS) void k_pass_scatter_packed(Loader ld, PassParams pp, uint2* out) {
    __shared__ uint2    s_stage[PT_TILE];
    __shared__ uint32_t s_cnt[PT_MAXF];
    __shared__ uint32_t s_base[PT_MAXF];
    __shared__ uint32_t s_delta[PT_MAXF];
    __shared__ uint32_t s_wsum[PT_THREADS / 64];
    uint32_t            seg, begin, end;
    if (!group_range(pp, blockIdx.x, seg, begin, end)) return;
    const uint32_t F = 1u << pp.fanout_log2, mask = F - 1u;

    uint32_t run = 0;  // thread d: where digit d's run of the current tile starts in the output
    // the next tile's loads are issued as soon as this tile has been staged (its registers are
    // dead then) and fly during the copy-out
    uint32_t w[PT_ITEMS][2];
    uint32_t ok_raw = ld.template issue_tile<2>(begin, end, w);
    for (uint32_t base = begin; base < end; base += PT_TILE) {
        for (uint32_t d = threadIdx.x; d < F; d += PT_THREADS) s_cnt[d] = 0;
        lds_barrier();

        uint32_t       dr[PT_ITEMS];  // digit << 16 | rank, 0xffffffff = no tuple
        const uint32_t ok = ld.template finish_tile<2>(base, end, ok_raw, w);  // (source tuples: NULL drop + hashing)
#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j) {
            dr[j] = 0xffffffffu;
            if ((ok >> j) & 1u) {
                uint32_t d = pass_digit(pp, w[j][0], mask);
                uint32_t r = atomicAdd(&s_cnt[d], 1u);
                dr[j] = (d << 16) | r;
            }
        }
        lds_barrier();

        uint32_t c = threadIdx.x < F ? s_cnt[threadIdx.x] : 0u;
        if (c)
            run = atomicAdd(&pp.cursor[(((size_t)seg * F + threadIdx.x) << pp.xcd_log2) |
                                       (blockIdx.x & ((1u << pp.xcd_log2) - 1u))], c);
        uint32_t total;
        uint32_t ex = block_excl_scan(c, s_wsum, total);
        if (threadIdx.x < F) s_base[threadIdx.x] = ex;
        lds_barrier();

#pragma unroll
        for (int j = 0; j < PT_ITEMS; ++j)
            if (dr[j] != 0xffffffffu)
                s_stage[s_base[dr[j] >> 16] + (dr[j] & 0xffffu)] = make_uint2(w[j][0], w[j][1]);
        if (base + PT_TILE < end) ok_raw = ld.template issue_tile<2>(base + PT_TILE, end, w);
        if (threadIdx.x < F) s_delta[threadIdx.x] = run - ex;  // global index = delta + LDS position
        lds_barrier();

#pragma unroll
        for (int k = 0; k < PT_ITEMS; ++k) {
            const uint32_t i = k * PT_THREADS + threadIdx.x;
            if (i < total) {
                const uint2    v = s_stage[i];
                const uint32_t g = s_delta[pass_digit(pp, v.x, mask)] + i;
                if constexpr (BLK_OUT) {  // (see BlockedLoader)
                    uint32_t* o = reinterpret_cast<uint32_t*>(out) + BlockedLoader::key_at(g);
                    o[0] = v.x;
                    o[BlockedLoader::BLK] = v.y;
                } else {
                    out[g] = v;
                }
                if (pp.side_out) pp.side_out[g] = (uint16_t)((v.x >> pp.next_shift) & pp.next_mask);
            }
        }
        // no barrier: the next tile passes two barriers before it overwrites s_stage / s_delta
    }
}

// ============================================================= heavy task list
// Probe partitions above JN_HEAVY tuples are cut into tasks so that a skewed
// (Zipf) probe side does not serialise on one workgroup; each task rebuilds the
// (small) build table of its partition.
__global__ void k_heavy_tasks(const uint32_t* offR, const uint32_t* offS, uint32_t NP,
                              uint32_t* tasks, uint32_t* n_heavy, uint32_t max_tasks) {
    uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= NP) return;
    uint32_t sb = offS[q], se = offS[q + 1];
    uint32_t len = se - sb;
    if (len <= JN_HEAVY || offR[q + 1] == offR[q]) return;
    uint32_t nt = (len + JN_HEAVY - 1) / JN_HEAVY;
    uint32_t b = atomicAdd(n_heavy, nt);
    for (uint32_t t = 0; t < nt; ++t) {
        uint32_t k = b + t;
        if (k >= max_tasks) break;  // cannot happen: max_tasks >= 2*|S|/JN_HEAVY + 1
        tasks[3 * k + 0] = q;
        // bounds in 64 bits: a partition ending within JN_HEAVY of 2^32 must not wrap
        tasks[3 * k + 1] = (uint32_t)((uint64_t)sb + (uint64_t)t * JN_HEAVY);
        tasks[3 * k + 2] = (uint32_t)min((uint64_t)se, (uint64_t)sb + (uint64_t)(t + 1) * JN_HEAVY);
    }
}

// ========================================================== K5/K6 build + probe
// Replaces the per-bucket table of the reference (src/execute.cpp:203-248):
//   build  — bucketised table in LDS: a tuple takes the next free slot of its 4-slot home
//            bucket (one LDS counter atomic), a full bucket overflows into the next one;
//            duplicate build keys simply occupy further slots;
//   probe  — one 16-byte read per bucket, four register compares; a bucket whose last slot
//            is EMPTY ends the walk; every equal key is a match (so duplicates multiply,
//            reference :232-243);
//   emit   — ballot/mbcnt offsets inside the wave, one global atomic per 4096
//            probe tuples reserves the output rows; lanes of a wave then write
//            consecutive rows of each output stream (coalesced).
// Bucket index uses hash bits ABOVE the radix bits (the reference reuses the low
// bits for both, SURVEY.md §3.2 — not copied).  EMPTY is a word whose radix bits
// differ from the partition's, so it cannot collide with a stored hashed key.
// A build partition larger than JN_RMAX is processed in table-sized chunks
// (block nested loop), which keeps any duplicate-heavy input correct.
// OM (output mode) fixes the stream layout at compile time for the two hot shapes, so the
// emit code is straight-line stores instead of a per-store mode switch:
//   OM_PAGED32  key/bc/pc are all INT32 Page images (root of a plan, BASELINE config)
//   OM_DENSE32  all present streams are dense 32-bit arrays (inner joins, row-index carries)
//   OM_GENERIC  anything else (64-bit streams, mixed layouts, no key stream)
//   OM_P32_64_64 INT32 key pages + INT64/FP64 pages for both carries (root of BASELINE config 3)
enum { OM_GENERIC = 0, OM_PAGED32 = 1, OM_DENSE32 = 2, OM_P32_64_64 = 3 };

// Diagnostic phase stamps (RJ_DIAG=1): thread 0 of every workgroup adds the cycles between
// consecutive stamps to jp.diag[phase].  Shares only — the stamps perturb the timing.
#define RJ_STAMP(PHASE)                                                         \
    do {                                                                        \
        if (jp.diag) {                                                          \
            unsigned long long _t = __builtin_amdgcn_s_memtime();               \
            if (threadIdx.x == 0) atomicAdd(&jp.diag[PHASE], _t - diag_t);      \
            diag_t = _t;                                                        \
        }                                                                       \
    } while (0)

// PK: bit 0 = the build side, bit 1 = the probe side is a packed {hashed key, carry} array
// TG: "tagged" table for one key word + a two-word build carry when the plan has >= 14 radix
//     bits.  All hashed keys of a partition share their low radix bits, so the remaining
//     (<= 18) high bits identify a key inside its partition: a slot is ONE word
//     {tag (19 bits) | build tuple index << 19} and the carries sit densely, indexed by build tuple.
//     Table = 32 KiB of slots + 32 KiB of carries (instead of three 32 KiB word arrays), so
//     two 512-thread workgroups share a CU and overlap their phases like the two-word join.
template <int KW, int CWR, int CWS, int OM, int PK, int TG>
__global__ __launch_bounds__(jn_threads(TG ? 2 : KW + CWR), jn_min_waves(TG ? 2 : KW + CWR)) void k_join(JoinParams jp) {
    // Tables of three or four word arrays (64-bit keys, two-word build carries) leave room for
    // ONE workgroup per CU; it then runs 1024 threads, so the CU holds the same 16 waves as with
    // two 512-thread workgroups.  (Fetching wide build carries from the partitioned arrays on
    // emit instead — a two-array table — cost 7 of 17 ms at 1 B rows: 2 random 4-byte reads per
    // match are bound by the request rate of the vector memory path.)
    static_assert(!TG || (KW == 1 && CWR == 2), "tagged table: one key word + two-word build carry");
    constexpr int      TH = jn_threads(TG ? 2 : KW + CWR);
    constexpr int      RPT = (JN_RMAX + TH - 1) / TH;  // build tuples per thread
    constexpr int      SPT = JN_SUB / TH;          // probe tuples per thread per sub-chunk
    constexpr int      SUB = JN_SUB;
    static_assert(RPT % 4 == 0 && SPT % 4 == 0, "tuples are loaded as 16-byte vectors");
    constexpr int      RW = KW + CWR;  // LDS table arrays (one per word)
    constexpr int      SW = KW + CWS;
    // Bucketised table: JN_CAP slots = JN_CAP/4 buckets of 4 consecutive slots.  A probe
    // reads a whole bucket with ONE 16-byte LDS read and compares in registers, an insert
    // takes its slot from ONE LDS counter atomic — so nearly every lane finishes in a single
    // iteration and the wave does not pay the longest linear-probing chain of its 64 lanes.
    // A full bucket (4th slot used) sends probe and insert on to the next bucket.
    __shared__ __attribute__((aligned(16))) uint32_t t_w[TG ? 1 : RW][JN_CAP];
    __shared__ __attribute__((aligned(16))) uint2    t_c2[TG ? JN_RMAX : 1];  // TG: carries by build tuple
    __shared__ __attribute__((aligned(16))) uint32_t t_cnt[JN_CAP / 4];
    __shared__ uint32_t s_wtot[TH / 64];
    __shared__ unsigned long long s_obase;

    const uint32_t     lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    constexpr uint32_t SMASK = JN_CAP - 1;
    constexpr uint32_t BMASK = JN_CAP / 4 - 1;

    // A workgroup joins PPW consecutive partitions (main pass) or one heavy task.
    constexpr int PPW = jn_ppw(TG ? 2 : KW + CWR);
    struct Task {
        uint32_t q, rbeg, rend, sbeg, send;
        bool     active;
    };
    // The first jp.heavy_grid workgroups of the launch take the heavy tasks (they start first:
    // with a skewed probe side they are the long pole), the others one partition each.
    const bool     heavy_wg = blockIdx.x < jp.heavy_grid;
    const uint32_t wg = heavy_wg ? blockIdx.x : blockIdx.x - jp.heavy_grid;
    auto get_task = [&](uint32_t k) {
        Task t{0, 0, 0, 0, 0, false};
        if (heavy_wg) {
            if (k == 0 && wg < *jp.n_heavy) {
                t.q = jp.heavy_tasks[3 * wg + 0];
                t.sbeg = jp.heavy_tasks[3 * wg + 1];
                t.send = jp.heavy_tasks[3 * wg + 2];
                t.rbeg = jp.offR[t.q];
                t.rend = jp.offR[t.q + 1];
                t.active = t.rbeg < t.rend && t.sbeg < t.send;
            }
            return t;
        }
        t.q = wg * PPW + k;
        if (k < (uint32_t)PPW && t.q < jp.NP) {
            t.rbeg = jp.offR[t.q];
            t.rend = jp.offR[t.q + 1];
            t.sbeg = jp.offS[t.q];
            t.send = jp.offS[t.q + 1];
            // partitions above JN_HEAVY probe tuples are split into tasks (k_heavy_tasks)
            t.active = t.rbeg < t.rend && t.sbeg < t.send && t.send - t.sbeg <= JN_HEAVY;
        }
        return t;
    };

    // item j of a thread is element ((j/4)*TH + tid)*4 + j%4: four consecutive
    // tuples per 16-byte load
    uint32_t rw[RPT][RW];
    uint32_t sw[SPT][SW];
    // item j of a thread holds element item_index(j): four consecutive tuples per 16-byte load
    // of a word array, two consecutive {key, carry} pairs per load of a packed array
    auto item_index = [&](int j, bool packed) -> uint32_t {
        return packed ? ((j / 2) * TH + threadIdx.x) * 2 + (j % 2)
                      : ((j / 4) * TH + threadIdx.x) * 4 + (j % 4);
    };
    constexpr bool packR = (PK & 1) != 0, packS = (PK & 2) != 0;
    constexpr bool aosR = (PK & 4) != 0, aosS = (PK & 8) != 0;
    static_assert(!aosR || (KW == 1 && CWR == 2), "12-byte build tuples: key + two-word carry");
    static_assert(!aosS || (KW == 1 && CWS == 2), "12-byte probe tuples: key + two-word carry");
    // four consecutive 12-byte tuples = three 16-byte loads (dword aligned)
    auto load_aos3 = [&](const uint32_t* base, uint32_t n, auto& regs, auto n_items) {
        constexpr int NI = decltype(n_items)::value;
#pragma unroll
        for (int v = 0; v < NI / 4; ++v) {
            const uint32_t  i0 = (v * TH + threadIdx.x) * 4;
            const uint32_t* p = base + (size_t)i0 * 3;
            if (i0 + 3 < n) {
                const u32x4a x = *reinterpret_cast<const u32x4a*>(p), y = *reinterpret_cast<const u32x4a*>(p + 4),
                             z = *reinterpret_cast<const u32x4a*>(p + 8);
                regs[4 * v + 0][0] = x[0];
                regs[4 * v + 0][1] = x[1];
                regs[4 * v + 0][2] = x[2];
                regs[4 * v + 1][0] = x[3];
                regs[4 * v + 1][1] = y[0];
                regs[4 * v + 1][2] = y[1];
                regs[4 * v + 2][0] = y[2];
                regs[4 * v + 2][1] = y[3];
                regs[4 * v + 2][2] = z[0];
                regs[4 * v + 3][0] = z[1];
                regs[4 * v + 3][1] = z[2];
                regs[4 * v + 3][2] = z[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int a = 0; a < 3; ++a) regs[4 * v + e][a] = i0 + e < n ? p[e * 3 + a] : 0u;
            }
        }
    };
    static_assert(!packR || (KW == 1 && CWR == 1), "packed build side: key + one carry word");
    static_assert(!packS || (KW == 1 && CWS == 1), "packed probe side: key + one carry word");
    auto load_build = [&](uint32_t rc, uint32_t rn) {
        constexpr int LW = RW;
        if constexpr (aosR) {
            load_aos3(jp.R.w[0] + (size_t)rc * 3, rn, rw, std::integral_constant<int, RPT>{});
            return;
        }
        if constexpr (packR) {
            {
                const uint2* rp = reinterpret_cast<const uint2*>(jp.R.w[0]) + rc;
#pragma unroll
                for (int v = 0; v < RPT / 2; ++v) {
                    const uint32_t i0 = (v * TH + threadIdx.x) * 2;
                    if (i0 + 1 < rn) {
                        u32x4a x = *reinterpret_cast<const u32x4a*>(rp + i0);
                        rw[2 * v][0] = x[0];
                        rw[2 * v][1] = x[1];
                        rw[2 * v + 1][0] = x[2];
                        rw[2 * v + 1][1] = x[3];
                    } else {
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            uint2 t = i0 + e < rn ? rp[i0 + e] : make_uint2(0u, 0u);
                            rw[2 * v + e][0] = t.x;
                            rw[2 * v + e][1] = t.y;
                        }
                    }
                }
                return;
            }
        }
        constexpr int NA = CWR >= 2 ? KW + CWR - 2 : LW;  // plain word arrays; the last two carry words are a pair array
#pragma unroll
        for (int v = 0; v < RPT / 4; ++v) {
            const uint32_t i0 = (v * TH + threadIdx.x) * 4;
            if (i0 + 3 < rn) {
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    u32x4a x = *reinterpret_cast<const u32x4a*>(jp.R.w[a] + rc + i0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) rw[4 * v + e][a] = x[e];
                }
                if constexpr (CWR >= 2) {
                    const uint32_t* p2 = jp.R.w[NA] + ((size_t)rc + i0) * 2;
                    u32x4a x = *reinterpret_cast<const u32x4a*>(p2), y = *reinterpret_cast<const u32x4a*>(p2 + 4);
                    rw[4 * v + 0][NA] = x[0];
                    rw[4 * v + 0][NA + 1] = x[1];
                    rw[4 * v + 1][NA] = x[2];
                    rw[4 * v + 1][NA + 1] = x[3];
                    rw[4 * v + 2][NA] = y[0];
                    rw[4 * v + 2][NA + 1] = y[1];
                    rw[4 * v + 3][NA] = y[2];
                    rw[4 * v + 3][NA + 1] = y[3];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int a = 0; a < NA; ++a)
                        rw[4 * v + e][a] = i0 + e < rn ? jp.R.w[a][rc + i0 + e] : 0u;
                    if constexpr (CWR >= 2) {
                        uint2 t = i0 + e < rn ? reinterpret_cast<const uint2*>(jp.R.w[NA])[rc + i0 + e]
                                              : make_uint2(0u, 0u);
                        rw[4 * v + e][NA] = t.x;
                        rw[4 * v + e][NA + 1] = t.y;
                    }
                }
            }
        }
    };
    auto load_probe = [&](uint32_t sc, uint32_t sn) {
        if constexpr (aosS) {
            load_aos3(jp.S.w[0] + (size_t)sc * 3, sn, sw, std::integral_constant<int, SPT>{});
            return;
        }
        if constexpr (packS) {
            {
                const uint2* sp = reinterpret_cast<const uint2*>(jp.S.w[0]) + sc;
#pragma unroll
                for (int v = 0; v < SPT / 2; ++v) {
                    const uint32_t i0 = (v * TH + threadIdx.x) * 2;
                    if (i0 + 1 < sn) {
                        u32x4a x = *reinterpret_cast<const u32x4a*>(sp + i0);
                        sw[2 * v][0] = x[0];
                        sw[2 * v][1] = x[1];
                        sw[2 * v + 1][0] = x[2];
                        sw[2 * v + 1][1] = x[3];
                    } else {
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            uint2 t = i0 + e < sn ? sp[i0 + e] : make_uint2(0u, 0u);
                            sw[2 * v + e][0] = t.x;
                            sw[2 * v + e][1] = t.y;
                        }
                    }
                }
                return;
            }
        }
        constexpr int NA = CWS >= 2 ? KW + CWS - 2 : SW;  // plain word arrays; the last two carry words are a pair array
#pragma unroll
        for (int v = 0; v < SPT / 4; ++v) {
            const uint32_t i0 = (v * TH + threadIdx.x) * 4;
            if (i0 + 3 < sn) {
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    u32x4a x = *reinterpret_cast<const u32x4a*>(jp.S.w[a] + sc + i0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) sw[4 * v + e][a] = x[e];
                }
                if constexpr (CWS >= 2) {
                    const uint32_t* p2 = jp.S.w[NA] + ((size_t)sc + i0) * 2;
                    u32x4a x = *reinterpret_cast<const u32x4a*>(p2), y = *reinterpret_cast<const u32x4a*>(p2 + 4);
                    sw[4 * v + 0][NA] = x[0];
                    sw[4 * v + 0][NA + 1] = x[1];
                    sw[4 * v + 1][NA] = x[2];
                    sw[4 * v + 1][NA + 1] = x[3];
                    sw[4 * v + 2][NA] = y[0];
                    sw[4 * v + 2][NA + 1] = y[1];
                    sw[4 * v + 3][NA] = y[2];
                    sw[4 * v + 3][NA + 1] = y[3];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int a = 0; a < NA; ++a)
                        sw[4 * v + e][a] = i0 + e < sn ? jp.S.w[a][sc + i0 + e] : 0u;
                    if constexpr (CWS >= 2) {
                        uint2 t = i0 + e < sn ? reinterpret_cast<const uint2*>(jp.S.w[NA])[sc + i0 + e]
                                              : make_uint2(0u, 0u);
                        sw[4 * v + e][NA] = t.x;
                        sw[4 * v + e][NA + 1] = t.y;
                    }
                }
            }
        }
    };
    // one output row: key + build carry + probe carry
    auto emit_row = [&](uint64_t row, uint32_t klo, uint32_t khi, uint32_t b0, uint32_t b1,
                        uint32_t p0, uint32_t p1, uint32_t b2 = 0u, uint32_t p2 = 0u) {
        if constexpr (OM == OM_PAGED32) {
            // one page/slot computation serves all three streams
            uint32_t r = (uint32_t)row, p = r / ROWS32;
            size_t   off = (size_t)p * PAGE_BYTES + HDR32 + (r - p * ROWS32) * 4u;
            *reinterpret_cast<uint32_t*>(jp.key.base + off) = klo;
            if constexpr (CWR >= 1) *reinterpret_cast<uint32_t*>(jp.bc.base + off) = b0;
            if constexpr (CWS >= 1) *reinterpret_cast<uint32_t*>(jp.pc.base + off) = p0;
        } else if constexpr (OM == OM_P32_64_64) {
            // one page/slot computation serves both 64-bit streams
            const uint32_t r = (uint32_t)row, p4 = r / ROWS32, p8 = r / ROWS64;
            *reinterpret_cast<uint32_t*>(jp.key.base + (size_t)p4 * PAGE_BYTES + HDR32 + (r - p4 * ROWS32) * 4u) = klo;
            const size_t off8 = (size_t)p8 * PAGE_BYTES + HDR64 + (r - p8 * ROWS64) * 8u;
            *reinterpret_cast<uint2*>(jp.bc.base + off8) = make_uint2(b0, b1);
            *reinterpret_cast<uint2*>(jp.pc.base + off8) = make_uint2(p0, p1);
        } else if constexpr (OM == OM_DENSE32) {
            reinterpret_cast<uint32_t*>(jp.key.base)[row] = klo;
            if constexpr (CWR >= 1) reinterpret_cast<uint32_t*>(jp.bc.base)[row] = b0;
            if constexpr (CWS >= 1) reinterpret_cast<uint32_t*>(jp.pc.base)[row] = p0;
        } else {
            stream_store(jp.key, row, klo, khi);
            if constexpr (CWR >= 1) stream_store(jp.bc, row, b0, b1, b2);
            if constexpr (CWS >= 1) stream_store(jp.pc, row, p0, p1, p2);
        }
    };

    unsigned long long diag_t = jp.diag ? __builtin_amdgcn_s_memtime() : 0ull;
    const uint32_t     n_tasks = heavy_wg ? 1u : (uint32_t)PPW;
    Task               cur = get_task(0);
    // haveR / haveS: the first build chunk / first probe sub-chunk of `cur` already sit in
    // rw / sw (prefetched while the previous partition was being probed / before its build)
    bool haveR = false, haveS = false;
    for (uint32_t k = 0; k < n_tasks; ++k) {
        const Task nxt = get_task(k + 1);
        if (!cur.active) {
            cur = nxt;
            haveR = haveS = false;
            continue;
        }
        if (!haveR) load_build(cur.rbeg, min((uint32_t)JN_RMAX, cur.rend - cur.rbeg));
        // TGLATE (experiment): the tagged variant loads its first probe sub-chunk only after the
        // build, when the build tuples' registers are free (peak live registers: 24 + 24 words)
#ifndef RJ_TG_LATE
#define RJ_TG_LATE 0
#endif
        constexpr bool late_probe = TG && RJ_TG_LATE;
        if (!haveS && !late_probe) load_probe(cur.sbeg, min((uint32_t)SUB, cur.send - cur.sbeg));
        uint32_t sw_pos = late_probe ? 0xffffffffu : cur.sbeg;  // which probe sub-chunk sw holds
        RJ_STAMP(0);  // loads issued

        // The low radix bits every hashed key of partition q shares, rebuilt from q
        // (q = ((d1*F2)+d2)*F3+d3, hash low bits = d1 | d2<<b1 | d3<<(b1+b2)); EMPTY differs
        // from them in bit 0, so no stored key can equal it.
        uint32_t qbits = 0;
        {
            uint32_t rem = cur.q, sh = jp.radix_bits;
            for (int p = (int)jp.n_pass - 1; p >= 0; --p) {
                uint32_t b = jp.pass_bits[p];
                sh -= b;
                qbits |= (rem & ((1u << b) - 1u)) << sh;
                rem >>= b;
            }
        }
        // TG: no build index reaches 0x1fff (a table holds JN_RMAX = 4096 tuples), so no slot equals
        // the all-ones word, and with >= 14 radix bits no tag (<= 18 bits) equals its 19 tag bits
        constexpr uint32_t TGB = 19, TGM = (1u << TGB) - 1u;
        static_assert(JN_RMAX <= (1u << (32 - TGB)) - 1u, "build index field of a tagged slot");
        const uint32_t EMPTY = TG ? 0xffffffffu : (qbits ^ 1u);

        for (uint32_t rc = cur.rbeg; rc < cur.rend; rc += JN_RMAX) {
            const uint32_t rn = min((uint32_t)JN_RMAX, cur.rend - rc);
            if (rc != cur.rbeg) load_build(rc, rn);
            {  // clear the key array, 16 bytes per store
                const uint4 e4 = make_uint4(EMPTY, EMPTY, EMPTY, EMPTY);
                uint4*      t4 = reinterpret_cast<uint4*>(&t_w[0][0]);
                for (uint32_t i = threadIdx.x; i < JN_CAP / 4; i += TH) t4[i] = e4;
                uint4* c4 = reinterpret_cast<uint4*>(&t_cnt[0]);
                for (uint32_t i = threadIdx.x; i < JN_CAP / 16; i += TH)
                    c4[i] = make_uint4(0, 0, 0, 0);
            }
            lds_barrier();
            RJ_STAMP(1);  // table cleared
            // ---- build
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                uint32_t i = item_index(j, packR);
                if (i < rn) {
                    uint32_t b = (rw[j][0] >> jp.radix_bits) & BMASK;
                    if constexpr (TG) t_c2[i] = make_uint2(rw[j][1], rw[j][2]);
                    while (true) {
                        uint32_t pos = atomicAdd(&t_cnt[b], 1u);
                        if (pos < 4) {
                            const uint32_t slot = b * 4 + pos;
                            if constexpr (TG) {
                                t_w[0][slot] = (rw[j][0] >> jp.radix_bits) | (i << TGB);
                            } else {
#pragma unroll
                                for (int a = 0; a < RW; ++a) t_w[a][slot] = rw[j][a];
                            }
                            break;
                        }
                        b = (b + 1) & BMASK;  // bucket full: overflow to the next one
                    }
                }
            }
            lds_barrier();
            RJ_STAMP(2);  // built
            // rw is dead now: start the NEXT partition's build loads behind this probe
            const bool last_chunk = rc + JN_RMAX >= cur.rend;
            if (last_chunk && nxt.active)
                load_build(nxt.rbeg, min((uint32_t)JN_RMAX, nxt.rend - nxt.rbeg));

            // ---- probe, SUB tuples at a time
            for (uint32_t sc = cur.sbeg; sc < cur.send; sc += SUB) {
                const uint32_t sn = min((uint32_t)SUB, cur.send - sc);
                if (sw_pos != sc) {
                    load_probe(sc, sn);
                    sw_pos = sc;
                }
                uint32_t m[SPT], f[SPT];
                // count matches, remember the first matching slot
#pragma unroll
                for (int j = 0; j < SPT; ++j) {
                    uint32_t i = item_index(j, packS);
                    m[j] = 0;
                    f[j] = 0;
                    if (i < sn) {
                        uint32_t b = (sw[j][0] >> jp.radix_bits) & BMASK;
                        while (true) {
                            const uint4 kv = *reinterpret_cast<const uint4*>(&t_w[0][b * 4]);
                            if constexpr (TG) {
                                const uint32_t tag = sw[j][0] >> jp.radix_bits;
                                const uint32_t eq = (uint32_t)((kv.x & TGM) == tag) | ((uint32_t)((kv.y & TGM) == tag) << 1) |
                                                    ((uint32_t)((kv.z & TGM) == tag) << 2) | ((uint32_t)((kv.w & TGM) == tag) << 3);
                                if (eq) {
                                    // f = build tuple index of the FIRST match (its carries: t_c2[f])
                                    if (m[j] == 0) f[j] = ((eq & 1u) ? kv.x : (eq & 2u) ? kv.y : (eq & 4u) ? kv.z : kv.w) >> TGB;
                                    m[j] += (uint32_t)__popc(eq);
                                }
                                if (kv.w == EMPTY) break;
                                b = (b + 1) & BMASK;
                                continue;
                            }
                            uint32_t    eq = (uint32_t)(kv.x == sw[j][0]) | ((uint32_t)(kv.y == sw[j][0]) << 1) |
                                          ((uint32_t)(kv.z == sw[j][0]) << 2) | ((uint32_t)(kv.w == sw[j][0]) << 3);
                            if (KW == 2 && eq) {
                                const uint4 hv = *reinterpret_cast<const uint4*>(&t_w[KW - 1][b * 4]);
                                eq &= (uint32_t)(hv.x == sw[j][KW - 1]) | ((uint32_t)(hv.y == sw[j][KW - 1]) << 1) |
                                      ((uint32_t)(hv.z == sw[j][KW - 1]) << 2) | ((uint32_t)(hv.w == sw[j][KW - 1]) << 3);
                            }
                            if (eq) {
                                if (m[j] == 0) f[j] = b * 4 + (uint32_t)__builtin_ctz(eq);
                                m[j] += (uint32_t)__popc(eq);
                            }
                            if (kv.w == EMPTY) break;  // bucket not full: nothing overflowed
                            b = (b + 1) & BMASK;
                        }
                    }
                }
                RJ_STAMP(3);  // counted
                // offsets inside the wave: ballot + mbcnt when every lane has <= 1 match
                // (the PK-FK case), shuffle scan otherwise
                uint32_t pre[SPT];
                uint32_t wave_total = 0;
#pragma unroll
                for (int j = 0; j < SPT; ++j) {
                    uint32_t tot;
                    if (__ballot(m[j] > 1) == 0) {
                        uint64_t mk = __ballot(m[j] == 1);
                        pre[j] = lane_prefix(mk);
                        tot = (uint32_t)__popcll(mk);
                    } else {
                        uint32_t incl = m[j];
#pragma unroll
                        for (int off = 1; off < 64; off <<= 1) {
                            uint32_t t = __shfl_up(incl, off);
                            if (lane >= (uint32_t)off) incl += t;
                        }
                        pre[j] = incl - m[j];
                        tot = __shfl(incl, 63);
                    }
                    pre[j] += wave_total;
                    wave_total += tot;
                }
                if (lane == 0) s_wtot[wid] = wave_total;
                lds_barrier();
                RJ_STAMP(4);  // wave prefixes + barrier
                if (threadIdx.x == 0) {
                    uint32_t tot = 0;
                    for (int w = 0; w < TH / 64; ++w) tot += s_wtot[w];
                    s_obase = tot ? atomicAdd(jp.out_cursor, (unsigned long long)tot) : 0ull;
                }
                lds_barrier();
                RJ_STAMP(5);  // output reservation
                const uint64_t gbase = s_obase;
                uint64_t       obase = gbase;
                uint32_t       block_total = 0;
                for (uint32_t w = 0; w < TH / 64; ++w) {
                    uint32_t t = s_wtot[w];
                    if (w < wid) obase += t;
                    block_total += t;
                }
                // rows beyond the stream capacity are counted but not written; the host
                // re-runs the join with exact-size buffers (out_cursor = rows needed)
                const bool fits = gbase + block_total <= jp.out_cap;
                if (fits) {
                    // the build carry of a tuple's FIRST match sits at the remembered slot
                    // f[j]: issue those reads for all tuples before the first store
                    // (TG reads them row by row instead: its registers are spoken for)
                    uint32_t c0[TG ? 1 : SPT], c1[TG ? 1 : SPT], c2[CWR == 3 ? SPT : 1];
                    if constexpr (!TG) {
#pragma unroll
                        for (int j = 0; j < SPT; ++j) {
                            c0[j] = 0;
                            c1[j] = 0;
                            if constexpr (CWR >= 1) c0[j] = m[j] ? t_w[KW][f[j]] : 0u;
                            if constexpr (CWR >= 2) c1[j] = m[j] ? t_w[KW + 1][f[j]] : 0u;
                            if constexpr (CWR == 3) c2[j] = m[j] ? t_w[KW + 2][f[j]] : 0u;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < SPT; ++j) {
                        if (m[j] == 0) continue;
                        uint64_t row = obase + pre[j];
                        uint32_t klo, khi = 0;
                        if (KW == 1) {
                            klo = unfmix32(sw[j][0]);
                        } else {
                            uint64_t k64 =
                                unfmix64((uint64_t)sw[j][0] | ((uint64_t)sw[j][KW - 1] << 32));
                            klo = (uint32_t)k64;
                            khi = (uint32_t)(k64 >> 32);
                        }
                        const uint32_t p0 = CWS >= 1 ? sw[j][KW < SW ? KW : 0] : 0u;
                        const uint32_t p1 = CWS >= 2 ? sw[j][KW + 1 < SW ? KW + 1 : 0] : 0u;
                        const uint32_t p2 = CWS == 3 ? sw[j][SW - 1] : 0u;
                        if constexpr (TG) {
                            const uint2 c = t_c2[f[j]];
                            emit_row(row, klo, khi, c.x, c.y, p0, p1, 0u, p2);
                        } else {
                            emit_row(row, klo, khi, c0[j], c1[j], p0, p1, CWR == 3 ? c2[CWR == 3 ? j : 0] : 0u, p2);
                        }
                        if constexpr (TG) {
                            // duplicates of the build key (rare): walk the buckets again from the
                            // home bucket and emit every match but the first
                            if (m[j] > 1) {
                                const uint32_t tag = sw[j][0] >> jp.radix_bits;
                                uint32_t       left = m[j], b = tag & BMASK;
                                while (left) {
                                    const uint4 kv = *reinterpret_cast<const uint4*>(&t_w[0][b * 4]);
                                    const uint32_t sv[4] = {kv.x, kv.y, kv.z, kv.w};
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        if ((sv[e] & TGM) == tag && left) {
                                            if (left != m[j]) {  // the first match went out above
                                                ++row;
                                                const uint2 c = t_c2[sv[e] >> TGB];
                                                emit_row(row, klo, khi, c.x, c.y, p0, p1, 0u, p2);
                                            }
                                            --left;
                                        }
                                    }
                                    b = (b + 1) & BMASK;
                                }
                            }
                            continue;
                        }
                        // duplicates of the build key occupy further slots of the same run
                        uint32_t left = m[j] - 1;
                        uint32_t slot = (f[j] + 1) & SMASK;
                        while (left) {
                            bool eq = t_w[0][slot] == sw[j][0];
                            if (KW == 2) eq = eq && t_w[KW - 1][slot] == sw[j][KW - 1];
                            if (eq) {
                                ++row;
                                uint32_t b0 = 0, b1 = 0, b2 = 0;
                                if constexpr (CWR >= 1) b0 = t_w[KW][slot];
                                if constexpr (CWR >= 2) b1 = t_w[KW + 1][slot];
                                if constexpr (CWR == 3) b2 = t_w[KW + 2][slot];
                                emit_row(row, klo, khi, b0, b1, p0, p1, b2, p2);
                                --left;
                            }
                            slot = (slot + 1) & SMASK;
                        }
                    }
                }
                RJ_STAMP(6);  // emitted
                lds_barrier();  // s_wtot / s_obase are reused by the next sub-chunk
            }
            lds_barrier();  // table is cleared for the next build chunk
        }
        // sw is dead now: start the next partition's probe loads behind its table build
        if (nxt.active) load_probe(nxt.sbeg, min((uint32_t)SUB, nxt.send - nxt.sbeg));
        haveR = haveS = nxt.active;
        cur = nxt;
    }
}

// ============================================================ broadcast join
// A build side of at most JN_RMAX rows (the filtered dimension tables of the JOB plans: a
// handful of rows against millions) needs no partitioning: every workgroup builds the SAME
// LDS table straight from the build child's columns and streams a slice of the probe
// child's columns past it — page decode, NULL-key drop and hashing happen on the way, the
// probe side is read exactly once and nothing is scattered.  (The reference counterpart is
// still hash_join_omp, src/execute.cpp:44-262, with num_buckets = 1.)
// The table keeps key words + the build ROW; build carries are fetched from the build
// columns on emit (a few KB: cache resident).  With no radix digit to derive an EMPTY
// sentinel from, a slot is valid when its index is below its bucket's insert counter.
template <int KW>
__device__ __forceinline__ bool src_key(const TupleSrc& s, uint32_t row, uint32_t& lo, uint32_t& hi) {
    bool ok = s.key.valid ? s.key.valid[row] != 0 : true;
    if constexpr (KW == 1) {
        uint32_t v = col_load32(s.key, row);
        lo = s.prehashed ? v : fmix32(v);
        hi = 0;
    } else {
        uint64_t k = col_load64(s.key, row);
        if (s.key_f64 && (k & 0x7ff0000000000000ull) == 0x7ff0000000000000ull &&
            (k & 0x000fffffffffffffull))
            ok = false;  // NaN never matches (see SrcLoader::hash_keys)
        uint64_t h = fmix64(k);
        lo = (uint32_t)h;
        hi = (uint32_t)(h >> 32);
    }
    return ok;
}
template <int CW>
__device__ __forceinline__ void src_carry(const TupleSrc& s, uint32_t row, uint32_t& c0, uint32_t& c1, uint32_t& c2) {
    c0 = c1 = c2 = 0;
    if constexpr (CW >= 1) {
        if (s.carry_mode == CARRY_WIDE) {
            if constexpr (CW >= 2) {
                if (s.wide == WIDE_64_32) {
                    const uint64_t v = col_load64(s.carry, row);
                    c0 = (uint32_t)v;
                    c1 = (uint32_t)(v >> 32);
                    c2 = col_load32(s.carry2, row);
                } else {
                    c0 = col_load32(s.carry, row);
                    c1 = col_load32(s.carry2, row);
                    if constexpr (CW == 3) c2 = col_load32(s.carry3, row);
                }
            }
        } else if (s.carry_mode == CARRY_ROWIDX) {
            c0 = row;
        } else if constexpr (CW == 1) {
            c0 = col_load32(s.carry, row);
        } else {
            uint64_t v = col_load64(s.carry, row);
            c0 = (uint32_t)v;
            c1 = (uint32_t)(v >> 32);
        }
    }
}

template <int KW, int CWR, int CWS>
__global__ __launch_bounds__(JN_THREADS) void k_join_bcast(BcastParams bp) {
    constexpr int RW = KW + 1;  // key words + build row
    __shared__ __attribute__((aligned(16))) uint32_t t_w[RW][JN_CAP];
    __shared__ __attribute__((aligned(16))) uint32_t t_cnt[JN_CAP / 4];
    __shared__ uint32_t s_wtot[JN_THREADS / 64];
    __shared__ unsigned long long s_obase;
    const uint32_t     lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    constexpr uint32_t BMASK = JN_CAP / 4 - 1;

    // ---- build (identical in every workgroup)
    for (uint32_t i = threadIdx.x; i < JN_CAP / 4; i += JN_THREADS) t_cnt[i] = 0;
    lds_barrier();
    for (uint32_t r = threadIdx.x; r < bp.R.n_rows; r += JN_THREADS) {
        uint32_t lo, hi;
        if (!src_key<KW>(bp.R, r, lo, hi)) continue;
        uint32_t b = lo & BMASK;
        while (true) {
            uint32_t pos = atomicAdd(&t_cnt[b], 1u);
            if (pos < 4) {
                const uint32_t slot = b * 4 + pos;
                t_w[0][slot] = lo;
                if constexpr (KW == 2) t_w[1][slot] = hi;
                t_w[KW][slot] = r;
                break;
            }
            b = (b + 1) & BMASK;
        }
    }
    lds_barrier();

    // matches of one probe key in bucket b: bit mask over its (valid) slots; `more` = the
    // bucket overflowed, the walk goes on in the next one
    auto match = [&](uint32_t b, uint32_t lo, uint32_t hi, bool& more) -> uint32_t {
        const uint32_t cnt = t_cnt[b];
        const uint4    kv = *reinterpret_cast<const uint4*>(&t_w[0][b * 4]);
        uint32_t       eq = (uint32_t)(kv.x == lo) | ((uint32_t)(kv.y == lo) << 1) |
                      ((uint32_t)(kv.z == lo) << 2) | ((uint32_t)(kv.w == lo) << 3);
        if constexpr (KW == 2) {
            const uint4 hv = *reinterpret_cast<const uint4*>(&t_w[1][b * 4]);
            eq &= (uint32_t)(hv.x == hi) | ((uint32_t)(hv.y == hi) << 1) |
                  ((uint32_t)(hv.z == hi) << 2) | ((uint32_t)(hv.w == hi) << 3);
        }
        more = cnt > 4;
        return eq & ((1u << min(cnt, 4u)) - 1u);
    };

    // ---- probe: chunks of JN_SUB rows, strided over the grid
    const uint32_t n = bp.S.n_rows;
    for (uint64_t base = (uint64_t)blockIdx.x * JN_SUB; base < n; base += (uint64_t)gridDim.x * JN_SUB) {
        uint32_t klo[JN_SPT], khi[JN_SPT], m[JN_SPT];
#pragma unroll
        for (int j = 0; j < JN_SPT; ++j) {
            const uint64_t row = base + (uint64_t)j * JN_THREADS + threadIdx.x;
            m[j] = 0;
            klo[j] = khi[j] = 0;
            if (row < n && src_key<KW>(bp.S, (uint32_t)row, klo[j], khi[j])) {
                uint32_t b = klo[j] & BMASK;
                bool     more;
                do {
                    m[j] += (uint32_t)__popc(match(b, klo[j], khi[j], more));
                    b = (b + 1) & BMASK;
                } while (more);
            }
        }
        // output offsets: wave prefix (ballot when every lane has <= 1 match), one global
        // reservation per chunk
        uint32_t pre[JN_SPT], wave_total = 0;
#pragma unroll
        for (int j = 0; j < JN_SPT; ++j) {
            uint32_t tot;
            if (__ballot(m[j] > 1) == 0) {
                uint64_t mk = __ballot(m[j] == 1);
                pre[j] = lane_prefix(mk);
                tot = (uint32_t)__popcll(mk);
            } else {
                uint32_t incl = m[j];
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    uint32_t t = __shfl_up(incl, off);
                    if (lane >= (uint32_t)off) incl += t;
                }
                pre[j] = incl - m[j];
                tot = __shfl(incl, 63);
            }
            pre[j] += wave_total;
            wave_total += tot;
        }
        if (lane == 0) s_wtot[wid] = wave_total;
        lds_barrier();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (int w = 0; w < JN_THREADS / 64; ++w) tot += s_wtot[w];
            s_obase = tot ? atomicAdd(bp.out_cursor, (unsigned long long)tot) : 0ull;
        }
        lds_barrier();
        const uint64_t gbase = s_obase;
        uint64_t       obase = gbase;
        uint32_t       block_total = 0;
        for (uint32_t w = 0; w < JN_THREADS / 64; ++w) {
            uint32_t t = s_wtot[w];
            if (w < wid) obase += t;
            block_total += t;
        }
        // rows beyond the stream capacity are counted but not written (the host re-runs the
        // join with exact-size buffers)
        if (gbase + block_total <= bp.out_cap) {
#pragma unroll
            for (int j = 0; j < JN_SPT; ++j) {
                if (m[j] == 0) continue;
                const uint32_t srow = (uint32_t)(base + (uint64_t)j * JN_THREADS + threadIdx.x);
                uint64_t       row = obase + pre[j];
                uint32_t       k0, k1 = 0;
                if (KW == 1) {
                    k0 = unfmix32(klo[j]);
                } else {
                    uint64_t k64 = unfmix64((uint64_t)klo[j] | ((uint64_t)khi[j] << 32));
                    k0 = (uint32_t)k64;
                    k1 = (uint32_t)(k64 >> 32);
                }
                uint32_t p0, p1, p2;
                src_carry<CWS>(bp.S, srow, p0, p1, p2);
                uint32_t b = klo[j] & BMASK;
                bool     more;
                do {
                    uint32_t eq = match(b, klo[j], khi[j], more);
                    while (eq) {
                        const uint32_t slot = b * 4 + (uint32_t)__builtin_ctz(eq);
                        eq &= eq - 1;
                        uint32_t b0, b1, b2;
                        src_carry<CWR>(bp.R, t_w[KW][slot], b0, b1, b2);
                        stream_store(bp.key, row, k0, k1);
                        if constexpr (CWR >= 1) stream_store(bp.bc, row, b0, b1, b2);
                        if constexpr (CWS >= 1) stream_store(bp.pc, row, p0, p1, p2);
                        ++row;
                    }
                    b = (b + 1) & BMASK;
                } while (more);
            }
        }
        lds_barrier();  // s_wtot / s_obase are reused by the next chunk
    }
}

// ================================================================== K7 gather
// Late materialisation: out[i] = column[idx[i]] (reference counterpart: the
// per-row `out.push_back(lrow[ci])`, src/execute.cpp:236-242).
template <int WIDTH>
__global__ __launch_bounds__(256) void k_gather(ColRef src, const uint32_t* idx, uint64_t n,
                                                OutStream dst, uint8_t* dst_valid) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r = idx ? idx[i] : (uint32_t)i;
    if (WIDTH == 4) {
        stream_store(dst, i, col_load32(src, r), 0u);
    } else {
        uint64_t v = col_load64(src, r);
        stream_store(dst, i, (uint32_t)v, (uint32_t)(v >> 32));
    }
    if (dst_valid) dst_valid[i] = src.valid ? src.valid[r] : (uint8_t)1;
}

// =========================================================== K8 page finishing
// Headers + all-ones validity bitmaps of result pages whose values were written
// in place by the probe / gather kernels (page layout: reference
// src/build_table.cpp:472-481).
__global__ __launch_bounds__(256) void k_finish_pages(uint8_t* pages, uint64_t n_rows,
                                                      uint32_t rows_full) {
    uint8_t*       page = pages + (size_t)blockIdx.x * PAGE_BYTES;
    const uint64_t first = (uint64_t)blockIdx.x * rows_full;
    const uint32_t nr = (uint32_t)min((uint64_t)rows_full, n_rows - first);
    const uint32_t nb = (nr + 7) / 8;
    if (threadIdx.x == 0) {
        reinterpret_cast<uint16_t*>(page)[0] = (uint16_t)nr;
        reinterpret_cast<uint16_t*>(page)[1] = (uint16_t)nr;
    }
    uint8_t* bm = page + PAGE_BYTES - nb;
    for (uint32_t k = threadIdx.x; k < nb; k += blockDim.x) {
        uint32_t bits = nr - k * 8u;
        bm[k] = bits >= 8 ? 0xff : (uint8_t)((1u << bits) - 1u);
    }
}

// Same, for the (up to three) streams the probe kernel wrote, with the row count still on
// the device: launched right behind the probe, before the host reads the count back.
struct FinishStreams {
    uint8_t* pages[3];
    uint32_t rows_full[3];
    uint32_t n;
};
__global__ __launch_bounds__(256) void k_finish_streams(FinishStreams fs,
                                                        const unsigned long long* n_rows_dev,
                                                        uint64_t cap_rows) {
    const uint64_t n_rows = *n_rows_dev;
    if (n_rows > cap_rows) return;  // overflow: the host re-runs the probe with larger buffers
    // one wave per page: the header and a bitmap of at most 248 bytes
    const uint32_t b = blockIdx.y, lane = threadIdx.x & 63u;
    const uint32_t rf = fs.rows_full[b];
    const uint64_t pg = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t first = pg * rf;
    if (first >= n_rows) return;
    uint8_t*       page = fs.pages[b] + (size_t)pg * PAGE_BYTES;
    const uint32_t nr = (uint32_t)min((uint64_t)rf, n_rows - first);
    const uint32_t nb = (nr + 7) / 8;
    if (lane == 0) *reinterpret_cast<uint32_t*>(page) = nr | (nr << 16);  // n_rows, n_nonnull
    uint8_t* bm = page + PAGE_BYTES - nb;
    if (nr == rf && (nr & 31u) == 0 && ((PAGE_BYTES - nb) & 3u) == 0) {
        // full INT32 page: 248 bytes of ones, dword aligned
        for (uint32_t k = lane; k < nb / 4; k += 64) reinterpret_cast<uint32_t*>(bm)[k] = 0xffffffffu;
        return;
    }
    if (nr == rf && rf == ROWS64) {
        // full INT64 / FP64 page: 1007 bits = 126 bytes from offset 8066 — one halfword, then 31
        // dwords (the last byte holds 7 bits)
        static_assert(ROWS64 == 1007 && ((PAGE_BYTES - 126 + 2) & 3u) == 0, "bitmap layout of a full 8-byte page");
        if (lane == 0) *reinterpret_cast<uint16_t*>(bm) = 0xffffu;
        if (lane < 31) reinterpret_cast<uint32_t*>(bm + 2)[lane] = lane == 30 ? 0x7fffffffu : 0xffffffffu;
        return;
    }
    for (uint32_t k = lane; k < nb; k += 64) {
        uint32_t bits = nr - k * 8u;
        bm[k] = bits >= 8 ? 0xff : (uint8_t)((1u << bits) - 1u);
    }
}

// Result pages of a column that carries NULLs: a fixed rows_full rows per page
// (any fill that satisfies the layout decodes to the same rows), values packed
// densely in row order, bitmap at the tail.
template <int WIDTH>
__global__ __launch_bounds__(256) void k_encode_nullable(const uint8_t* values, const uint8_t* valid,
                                                         uint64_t n_rows, uint8_t* pages) {
    __shared__ uint32_t s_w[4];
    constexpr uint32_t  RF = WIDTH == 4 ? ROWS32 : ROWS64;
    uint8_t*            page = pages + (size_t)blockIdx.x * PAGE_BYTES;
    const uint64_t      first = (uint64_t)blockIdx.x * RF;
    const uint32_t      nr = (uint32_t)min((uint64_t)RF, n_rows - first);
    const uint32_t      nb = (nr + 7) / 8;
    uint8_t*            out_vals = page + (WIDTH == 4 ? HDR32 : HDR64);
    uint8_t*            bm = page + PAGE_BYTES - nb;
    const uint32_t      lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    uint32_t            running = 0;
    for (uint32_t base = 0; base < nr; base += 256) {
        uint32_t i = base + threadIdx.x;
        bool     bit = i < nr ? valid[first + i] != 0 : false;
        uint64_t mask = __ballot(bit);
        uint32_t pre = lane_prefix(mask);
        if (lane == 0) s_w[wid] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t wpre = 0, tot = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            uint32_t c = s_w[k];
            if (k < wid) wpre += c;
            tot += c;
        }
        if (bit) {
            uint32_t vi = running + wpre + pre;
            if (WIDTH == 4)
                reinterpret_cast<uint32_t*>(out_vals)[vi] =
                    reinterpret_cast<const uint32_t*>(values)[first + i];
            else
                reinterpret_cast<uint64_t*>(out_vals)[vi] =
                    reinterpret_cast<const uint64_t*>(values)[first + i];
        }
        // bitmap bytes of this 256-row slab: the wave's ballot holds 8 of them
        if (lane < 8) {
            uint32_t byte_idx = (base >> 3) + wid * 8u + lane;
            if (byte_idx < nb) bm[byte_idx] = (uint8_t)(mask >> (lane * 8u));
        }
        running += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        reinterpret_cast<uint16_t*>(page)[0] = (uint16_t)nr;
        reinterpret_cast<uint16_t*>(page)[1] = (uint16_t)running;
    }
}

// ============================================================ wide carries
// Validity of up to three carried columns as ONE 32-bit word per row (bit c = column c is
// non-NULL): the word then travels like a 32-bit column of the wide carry.
__global__ __launch_bounds__(256) void k_pack_validity(const uint8_t* v0, const uint8_t* v1, const uint8_t* v2,
                                                       uint32_t n, uint32_t* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = (uint32_t)(!v0 || v0[i]) | ((uint32_t)(!v1 || v1[i]) << 1) | ((uint32_t)(!v2 || v2[i]) << 2);
}

// The records a join emitted for a wide carry (cw words per output row) -> one dense array per
// carried column, plus validity bytes for the columns that have them.  Sequential reads and
// writes: this replaces k_gather's random reads through a row-index stream.
__global__ __launch_bounds__(256) void k_split_records(SplitParams sp, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* r = sp.rec + i * sp.cw;
    const uint32_t  vw = sp.valid_word >= 0 ? r[sp.valid_word] : 0xffffffffu;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (c >= sp.n_cols) break;
        const uint32_t lo = r[sp.col[c].word];
        if (sp.col[c].paged)
            stream_store(OutStream{sp.col[c].out, sp.col[c].width == 8 ? ST_PAGED64 : ST_PAGED32, 0}, i, lo,
                         sp.col[c].width == 8 ? r[sp.col[c].word + 1] : 0u);
        else if (sp.col[c].width == 8)
            reinterpret_cast<uint64_t*>(sp.col[c].out)[i] = (uint64_t)lo | ((uint64_t)r[sp.col[c].word + 1] << 32);
        else
            reinterpret_cast<uint32_t*>(sp.col[c].out)[i] = lo;
        if (sp.col[c].valid) sp.col[c].valid[i] = (uint8_t)((vw >> sp.col[c].valid_bit) & 1u);
    }
}

// Test hook (RJ_DEBUG_SHARD_FAIL=4): holds a stream for `ticks` of the constant-rate wall clock
// (hipDeviceAttributeWallClockRate) — a rank whose stage A "never finishes", so that its peers'
// bounded waits can be seen to expire.  Every wave leaves on its own when the time is up.
__global__ void k_debug_stall(unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(127);
}

// ================================================================== launchers
// A rejected launch (LDS or launch-bounds mismatch of a tuning variant, wrong device) must not
// pass silently: the stream would "succeed" and the join return stale buffers with RJ_OK.
#define RJ_KLAUNCH(L, NAME, KERNEL, GRID, BLOCK, ...)                                          \
    do {                                                                                       \
        hipEvent_t _ev0 = nullptr, _ev1 = nullptr;                                             \
        if ((L).timed && (L).timed((L).self, NAME, &_ev0, &_ev1))                              \
            hipExtLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), 0, (L).stream, _ev0, _ev1,  \
                                  0, __VA_ARGS__);                                             \
        else                                                                                   \
            hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), 0, (L).stream, __VA_ARGS__);   \
        hipError_t _le = hipGetLastError();                                                    \
        if (_le != hipSuccess) launch_failed(NAME, hipGetErrorString(_le), false);             \
    } while (0)

void launch_page_headers(const Launch& L, const uint8_t* pages, uint32_t n_pages, uint32_t rows_full,
                         uint32_t* page_rows, unsigned long long* flags) {
    if (!n_pages) return;
    RJ_KLAUNCH(L, "page_headers", k_page_headers, std::min<uint32_t>((n_pages + 3) / 4, 8192u), 256, pages, n_pages,
               rows_full, page_rows, flags);
}

void launch_rows_beyond(const Launch& L, const uint8_t* pages, uint32_t n_pages,
                        const uint32_t* row_base, uint64_t num_rows, unsigned long long* flag) {
    if (!n_pages) return;
    RJ_KLAUNCH(L, "rows_beyond", k_rows_beyond, (n_pages + 3) / 4, 256, pages, n_pages, row_base,
               num_rows, flag);
}

void launch_decode_pages(const Launch& L, const uint8_t* pages, uint32_t n_pages, int width,
                         const uint32_t* row_base, uint64_t num_rows, uint8_t* values,
                         uint8_t* valid) {
    if (!n_pages) return;
    if (width == 4)
        RJ_KLAUNCH(L, "decode_pages", (k_decode_pages<4>), n_pages, 256, pages, row_base, num_rows,
                   values, valid);
    else
        RJ_KLAUNCH(L, "decode_pages", (k_decode_pages<8>), n_pages, 256, pages, row_base, num_rows,
                   values, valid);
}

void launch_scan_bins(const Launch& L, const uint32_t* in, uint32_t n, uint32_t* off,
                      uint32_t* cursor) {
    RJ_KLAUNCH(L, "scan_bins", (k_scan_bins<0>), 1, 1024, in, (const uint32_t*)nullptr, n, 1u, off, cursor);
}

void launch_scan_segments(const Launch& L, const uint32_t* hist, const uint32_t* seg_off,
                          uint32_t nseg, uint32_t F, uint32_t xcd_log2, uint32_t* off, uint32_t* cursor) {
    RJ_KLAUNCH(L, "scan_segments", k_scan_segments, nseg, PT_MAXF, hist, seg_off, nseg, F, xcd_log2, off,
               cursor);
}

void launch_group_table(const Launch& L, const uint32_t* seg_off, uint32_t nseg,
                        uint32_t group_tuples, uint32_t* grp_start, const uint32_t* seg_end) {
    RJ_KLAUNCH(L, "group_table", (k_scan_bins<1>), 1, 1024, seg_off, seg_end, nseg, group_tuples, grp_start,
               (uint32_t*)nullptr);
}

void launch_pass_hist_src(const Launch& L, const TupleSrc& src, int key_words, const PassParams& pp,
                          uint32_t n_groups) {
    if (!n_groups) return;
    if (key_words == 1) {
        SrcLoader<1, 0> ld{src};
        RJ_KLAUNCH(L, "pass1_hist", (k_pass_hist<SrcLoader<1, 0>>), n_groups, PT_THREADS, ld, pp);
    } else {
        SrcLoader<2, 0> ld{src};
        RJ_KLAUNCH(L, "pass1_hist", (k_pass_hist<SrcLoader<2, 0>>), n_groups, PT_THREADS, ld, pp);
    }
}

void launch_fine_hist_src(const Launch& L, const TupleSrc& src, int key_words, uint32_t shift,
                          uint32_t b1, uint32_t b2, uint32_t grid, uint32_t* fine, uint32_t xcd_tpg,
                          uint32_t* coarse_x) {
    if (!grid || !src.n_rows) return;
    if (key_words == 1) {
        SrcLoader<1, 0> ld{src};
        RJ_KLAUNCH(L, "pass1_hist", (k_fine_hist<SrcLoader<1, 0>>), grid, PT_THREADS, ld, src.n_rows,
                   shift, b1, b2, fine, xcd_tpg, coarse_x);
    } else {
        SrcLoader<2, 0> ld{src};
        RJ_KLAUNCH(L, "pass1_hist", (k_fine_hist<SrcLoader<2, 0>>), grid, PT_THREADS, ld, src.n_rows,
                   shift, b1, b2, fine, xcd_tpg, coarse_x);
    }
}

void launch_fine_hist_words(const Launch& L, const Words& in, bool packed, uint32_t n, uint32_t shift,
                            uint32_t b1, uint32_t b2, uint32_t grid, uint32_t* fine, uint32_t xcd_tpg,
                            uint32_t* coarse_x) {
    if (!grid || !n) return;
    if (packed) {
        PackedLoader ld{reinterpret_cast<const uint2*>(in.w[0])};
        RJ_KLAUNCH(L, "pass1_hist", (k_fine_hist<PackedLoader>), grid, PT_THREADS, ld, n, shift, b1, b2, fine,
                   xcd_tpg, coarse_x);
    } else {
        DenseLoader ld{in};
        RJ_KLAUNCH(L, "pass1_hist", (k_fine_hist<DenseLoader>), grid, PT_THREADS, ld, n, shift, b1, b2, fine,
                   xcd_tpg, coarse_x);
    }
}

void launch_scan_fine(const Launch& L, const uint32_t* fine, uint32_t F1, uint32_t F2,
                      uint32_t* off2, uint32_t* cursor2, uint32_t* off1, uint32_t* cursor1,
                      const uint32_t* coarse_x) {
    RJ_KLAUNCH(L, "scan_fine", k_scan_fine, F1, PT_MAXF, fine, F1, F2, off2, cursor2, off1, cursor1,
               coarse_x);
}

// (carries of two or more words: the last two travel as one array of 8-byte pairs)
template <int KW, int CW, int WIDE = WIDE_NONE>
static void scatter_src_t(const Launch& L, const TupleSrc& src, const PassParams& pp,
                          uint32_t n_groups, const Words& out) {
    SrcLoader<KW, CW, WIDE> ld{src};
    RJ_KLAUNCH(L, "pass1_scatter", (k_pass_scatter<KW + CW, SrcLoader<KW, CW, WIDE>, (CW >= 2 ? KW + CW - 2 : -1), false>),
               n_groups, PT_THREADS, ld, pp, out);
}

void launch_pass_scatter_src(const Launch& L, const TupleSrc& src, int key_words, int carry_words,
                             const PassParams& pp, uint32_t n_groups, const Words& out, bool aos3) {
    if (!n_groups) return;
    const int wide = src.carry_mode == CARRY_WIDE ? src.wide : WIDE_NONE;
    if (aos3) {
        if (key_words != 1 || carry_words != 2) launch_failed("pass1_scatter", "12-byte tuples need KW=1, CW=2", true);
        if (wide == WIDE_32S) {
            SrcLoader<1, 2, WIDE_32S> ld{src};
            RJ_KLAUNCH(L, "pass1_scatter", (k_pass_scatter<3, SrcLoader<1, 2, WIDE_32S>, 1, true>), n_groups, PT_THREADS, ld, pp, out);
            return;
        }
        SrcLoader<1, 2> ld{src};
        RJ_KLAUNCH(L, "pass1_scatter", (k_pass_scatter<3, SrcLoader<1, 2>, 1, true>), n_groups, PT_THREADS, ld, pp, out);
        return;
    }
    if (wide != WIDE_NONE) {
        switch (key_words * 100 + carry_words * 10 + wide) {
        case 121: scatter_src_t<1, 2, WIDE_32S>(L, src, pp, n_groups, out); break;
        case 131: scatter_src_t<1, 3, WIDE_32S>(L, src, pp, n_groups, out); break;
        case 132: scatter_src_t<1, 3, WIDE_64_32>(L, src, pp, n_groups, out); break;
        case 221: scatter_src_t<2, 2, WIDE_32S>(L, src, pp, n_groups, out); break;
        default: launch_failed("pass1_scatter", "no kernel for this wide carry layout", true);
        }
        return;
    }
    switch (key_words * 10 + carry_words) {
    case 10: scatter_src_t<1, 0>(L, src, pp, n_groups, out); break;
    case 11: scatter_src_t<1, 1>(L, src, pp, n_groups, out); break;
    case 12: scatter_src_t<1, 2>(L, src, pp, n_groups, out); break;
    case 20: scatter_src_t<2, 0>(L, src, pp, n_groups, out); break;
    case 21: scatter_src_t<2, 1>(L, src, pp, n_groups, out); break;
    case 22: scatter_src_t<2, 2>(L, src, pp, n_groups, out); break;
    default: launch_failed("pass1_scatter", "no kernel for this key/carry word count", true);
    }
}

void launch_pass_hist_dense(const Launch& L, const Words& in, const PassParams& pp,
                            uint32_t n_groups) {
    if (!n_groups) return;
    DenseLoader ld{in};
    RJ_KLAUNCH(L, "pass2_hist", (k_pass_hist<DenseLoader>), n_groups, PT_THREADS, ld, pp);
}

void launch_pass_scatter_dense(const Launch& L, const Words& in, int n_words, int pair_word,
                               const PassParams& pp, uint32_t n_groups, const Words& out, bool aos3) {
    if (!n_groups) return;
    if (aos3) {
        if (n_words != 3 || pair_word != 1) launch_failed("pass2_scatter", "12-byte tuples need KW=1, CW=2", true);
        RJ_KLAUNCH(L, "pass2_scatter", (k_pass_scatter<3, DenseLoaderT<1>, 1, true>), n_groups, PT_THREADS,
                   DenseLoaderT<1>{in}, pp, out);
        return;
    }
    switch (n_words * 10 + (pair_word < 0 ? 9 : pair_word)) {
    case 19:
        RJ_KLAUNCH(L, "pass2_scatter", (k_pass_scatter<1, DenseLoader, -1, false>), n_groups, PT_THREADS,
                   DenseLoader{in}, pp, out);
        break;
    case 29:
        RJ_KLAUNCH(L, "pass2_scatter", (k_pass_scatter<2, DenseLoader, -1, false>), n_groups, PT_THREADS,
                   DenseLoader{in}, pp, out);
        break;
    case 39:
        RJ_KLAUNCH(L, "pass2_scatter", (k_pass_scatter<3, DenseLoader, -1, false>), n_groups, PT_THREADS,
                   DenseLoader{in}, pp, out);
        break;
    case 31:  // key + carry pair
        RJ_KLAUNCH(L, "pass2_scatter", (k_pass_scatter<3, DenseLoaderT<1>, 1, false>), n_groups, PT_THREADS,
                   DenseLoaderT<1>{in}, pp, out);
        break;
    case 42:  // two key words + carry pair
        RJ_KLAUNCH(L, "pass2_scatter", (k_pass_scatter<4, DenseLoaderT<2>, 2, false>), n_groups, PT_THREADS,
                   DenseLoaderT<2>{in}, pp, out);
        break;
    default: launch_failed("pass2_scatter", "no kernel for this word layout", true);
    }
}

void launch_pass_hist_digits(const Launch& L, const uint16_t* digits, const PassParams& pp, uint32_t n_groups) {
    if (!n_groups) return;
    DigitLoader ld{digits, pp.shift};
    RJ_KLAUNCH(L, "pass2_hist", (k_pass_hist<DigitLoader>), n_groups, PT_THREADS, ld, pp);
}

void launch_pass_hist_aos3(const Launch& L, const uint32_t* in_tuples, const PassParams& pp, uint32_t n_groups) {
    if (!n_groups) return;
    RJ_KLAUNCH(L, "pass2_hist", (k_pass_hist<Aos3KeyLoader>), n_groups, PT_THREADS, Aos3KeyLoader{in_tuples}, pp);
}

void launch_pass_scatter_aos3(const Launch& L, const uint32_t* in_tuples, const PassParams& pp, uint32_t n_groups,
                              uint32_t* out_tuples) {
    if (!n_groups) return;
    Words out{};
    out.w[0] = out_tuples;
    RJ_KLAUNCH(L, "pass2_scatter", (k_pass_scatter<3, Aos3Loader, 1, true>), n_groups, PT_THREADS,
               Aos3Loader{in_tuples}, pp, out);
}

void launch_pass_hist_packed(const Launch& L, const uint32_t* in_pairs, const PassParams& pp,
                             uint32_t n_groups) {
    if (!n_groups) return;
    PackedLoader ld{reinterpret_cast<const uint2*>(in_pairs)};
    RJ_KLAUNCH(L, "pass2_hist", (k_pass_hist<PackedLoader>), n_groups, PT_THREADS, ld, pp);
}

void launch_pass_scatter_src_packed(const Launch& L, const TupleSrc& src, const PassParams& pp,
                                    uint32_t n_groups, uint32_t* out_pairs, bool blocked_out) {
    if (!n_groups) return;
    SrcLoader<1, 1> ld{src};
    if (blocked_out)
        RJ_KLAUNCH(L, "pass1_scatter", (k_pass_scatter_packed<SrcLoader<1, 1>, true>), n_groups, PT_THREADS, ld,
                   pp, reinterpret_cast<uint2*>(out_pairs));
    else
        RJ_KLAUNCH(L, "pass1_scatter", (k_pass_scatter_packed<SrcLoader<1, 1>>), n_groups, PT_THREADS, ld,
                   pp, reinterpret_cast<uint2*>(out_pairs));
}

// a later pass over BLOCKED pairs (see BlockedLoader); its output is blocked again, or — the last pass — packed
void launch_pass_hist_blocked(const Launch& L, const uint32_t* in_blocked, const PassParams& pp, uint32_t n_groups) {
    if (!n_groups) return;
    RJ_KLAUNCH(L, "pass2_hist", (k_pass_hist<BlockedLoader>), n_groups, PT_THREADS, BlockedLoader{in_blocked}, pp);
}
void launch_pass_scatter_blocked(const Launch& L, const uint32_t* in_blocked, const PassParams& pp, uint32_t n_groups,
                                 uint32_t* out_pairs, bool blocked_out) {
    if (!n_groups) return;
    BlockedLoader ld{in_blocked};
    if (blocked_out)
        RJ_KLAUNCH(L, "pass2_scatter", (k_pass_scatter_packed<BlockedLoader, true>), n_groups, PT_THREADS, ld, pp,
                   reinterpret_cast<uint2*>(out_pairs));
    else
        RJ_KLAUNCH(L, "pass2_scatter", (k_pass_scatter_packed<BlockedLoader>), n_groups, PT_THREADS, ld, pp,
                   reinterpret_cast<uint2*>(out_pairs));
}

void launch_pass_scatter_packed(const Launch& L, const uint32_t* in_pairs, const PassParams& pp,
                                uint32_t n_groups, uint32_t* out_pairs) {
    if (!n_groups) return;
    PackedLoader ld{reinterpret_cast<const uint2*>(in_pairs)};
    RJ_KLAUNCH(L, "pass2_scatter", (k_pass_scatter_packed<PackedLoader>), n_groups, PT_THREADS, ld, pp,
               reinterpret_cast<uint2*>(out_pairs));
}

void launch_heavy_tasks(const Launch& L, const uint32_t* offR, const uint32_t* offS, uint32_t NP,
                        uint32_t* tasks, uint32_t* n_heavy, uint32_t max_tasks) {
    RJ_KLAUNCH(L, "heavy_tasks", k_heavy_tasks, (NP + 255) / 256, 256, offR, offS, NP, tasks,
               n_heavy, max_tasks);
}

// tagged table (see k_join): one key word + two-word build carry, >= 14 radix bits
#ifndef RJ_TG_ENABLE
#define RJ_TG_ENABLE 1
#endif
static bool join_tagged(int key_words, int cw_build, const JoinParams& jp) {
    return RJ_TG_ENABLE && key_words == 1 && cw_build == 2 && jp.radix_bits >= 14 && jp.radix_bits <= 31;
}

template <int KW, int CWR, int CWS, int PK>
static void join_pk(const Launch& L, const JoinParams& jp, uint32_t grid) {
    const char* name = "join_build_probe";
    // (the tagged table also serves a PACKED probe side — key + one carry word, PK bit 1: the shape of
    // every join whose build side carries two words and whose probe side one; 1.45 -> 0.9 ms at 100 M rows)
    if constexpr (KW == 1 && CWR == 2 && (PK & 1) == 0) {
        const bool p366 = CWS == 2 && jp.key.mode == ST_PAGED32 && jp.bc.mode == ST_PAGED64 &&
                          jp.pc.mode == ST_PAGED64;
        if (join_tagged(KW, CWR, jp)) {
            if constexpr (CWS == 2) {
                if (p366) {
                    RJ_KLAUNCH(L, name, (k_join<KW, CWR, CWS, OM_P32_64_64, PK, 1>), grid, jn_threads(2), jp);
                    return;
                }
            }
            RJ_KLAUNCH(L, name, (k_join<KW, CWR, CWS, OM_GENERIC, PK, 1>), grid, jn_threads(2), jp);
            return;
        }
        if constexpr (CWS == 2) {
            if (p366) {
                RJ_KLAUNCH(L, name, (k_join<KW, CWR, CWS, OM_P32_64_64, PK, 0>), grid, jn_threads(KW + CWR), jp);
                return;
            }
        }
    }
    // pick the straight-line emit variant when the stream layout allows it
    int om = OM_GENERIC;
    if (KW == 1 && CWR <= 1 && CWS <= 1) {
        auto all = [&](int mode) {
            return jp.key.mode == mode && (CWR == 0 || jp.bc.mode == mode) &&
                   (CWS == 0 || jp.pc.mode == mode);
        };
        if (all(ST_PAGED32)) om = OM_PAGED32;
        if (all(ST_DENSE32)) om = OM_DENSE32;
    }
    if constexpr (KW == 1 && CWR <= 1 && CWS <= 1) {
        if (om == OM_PAGED32) {
            RJ_KLAUNCH(L, name, (k_join<KW, CWR, CWS, OM_PAGED32, PK, 0>), grid, jn_threads(KW + CWR), jp);
            return;
        }
        if (om == OM_DENSE32) {
            RJ_KLAUNCH(L, name, (k_join<KW, CWR, CWS, OM_DENSE32, PK, 0>), grid, jn_threads(KW + CWR), jp);
            return;
        }
    }
    RJ_KLAUNCH(L, name, (k_join<KW, CWR, CWS, OM_GENERIC, PK, 0>), grid, jn_threads(KW + CWR), jp);
}

// the packed / 12-byte variants exist only for the shapes that can have them
template <int KW, int CWR, int CWS>
static void join_t(const Launch& L, const JoinParams& jp, uint32_t grid) {
    constexpr int CAN = (KW == 1 && CWR == 1 ? 1 : 0) | (KW == 1 && CWS == 1 ? 2 : 0) |
                        (KW == 1 && CWR == 2 ? 4 : 0) | (KW == 1 && CWS == 2 ? 8 : 0);
    const int     pk = (jp.packR ? 1 : 0) | (jp.packS ? 2 : 0) | (jp.aosR ? 4 : 0) | (jp.aosS ? 8 : 0);
    if (pk & ~CAN) launch_failed("join_build_probe", "tuple layout flags do not fit the key/carry widths", true);
#define RJ_JOIN_PK(V)                 \
    if constexpr ((CAN & (V)) == (V)) \
        if (pk == (V)) return join_pk<KW, CWR, CWS, (V)>(L, jp, grid);
    RJ_JOIN_PK(1)
    RJ_JOIN_PK(2)
    RJ_JOIN_PK(3)
    RJ_JOIN_PK(4)
    RJ_JOIN_PK(8)
    RJ_JOIN_PK(12)
    RJ_JOIN_PK(6)
    RJ_JOIN_PK(9)
#undef RJ_JOIN_PK
    if (pk != 0) launch_failed("join_build_probe", "no kernel for this tuple layout", true);
    join_pk<KW, CWR, CWS, 0>(L, jp, grid);
}

uint32_t join_partitions_per_workgroup(int key_words, int cw_build, const JoinParams& jp) {
    return (uint32_t)jn_ppw(join_tagged(key_words, cw_build, jp) ? 2 : key_words + cw_build);
}

void launch_join(const Launch& L, int key_words, int cw_build, int cw_probe, const JoinParams& jp,
                 uint32_t grid) {
    if (!grid) return;
    switch (key_words * 100 + cw_build * 10 + cw_probe) {
    case 100: join_t<1, 0, 0>(L, jp, grid); break;
    case 101: join_t<1, 0, 1>(L, jp, grid); break;
    case 102: join_t<1, 0, 2>(L, jp, grid); break;
    case 110: join_t<1, 1, 0>(L, jp, grid); break;
    case 111: join_t<1, 1, 1>(L, jp, grid); break;
    case 112: join_t<1, 1, 2>(L, jp, grid); break;
    case 120: join_t<1, 2, 0>(L, jp, grid); break;
    case 121: join_t<1, 2, 1>(L, jp, grid); break;
    case 122: join_t<1, 2, 2>(L, jp, grid); break;
    case 103: join_t<1, 0, 3>(L, jp, grid); break;  // three-word (wide) carries
    case 113: join_t<1, 1, 3>(L, jp, grid); break;
    case 123: join_t<1, 2, 3>(L, jp, grid); break;
    case 130: join_t<1, 3, 0>(L, jp, grid); break;
    case 131: join_t<1, 3, 1>(L, jp, grid); break;
    case 132: join_t<1, 3, 2>(L, jp, grid); break;
    case 133: join_t<1, 3, 3>(L, jp, grid); break;
    case 200: join_t<2, 0, 0>(L, jp, grid); break;
    case 201: join_t<2, 0, 1>(L, jp, grid); break;
    case 202: join_t<2, 0, 2>(L, jp, grid); break;
    case 210: join_t<2, 1, 0>(L, jp, grid); break;
    case 211: join_t<2, 1, 1>(L, jp, grid); break;
    case 212: join_t<2, 1, 2>(L, jp, grid); break;
    case 220: join_t<2, 2, 0>(L, jp, grid); break;
    case 221: join_t<2, 2, 1>(L, jp, grid); break;
    case 222: join_t<2, 2, 2>(L, jp, grid); break;
    default: launch_failed("join_build_probe", "no kernel for this key/carry word count", true);
    }
}

template <int KW, int CWR, int CWS>
static void join_bcast_t(const Launch& L, const BcastParams& bp, uint32_t grid) {
    RJ_KLAUNCH(L, "join_broadcast", (k_join_bcast<KW, CWR, CWS>), grid, JN_THREADS, bp);
}

void launch_join_bcast(const Launch& L, int key_words, int cw_build, int cw_probe,
                       const BcastParams& bp, uint32_t grid) {
    if (!grid) return;
    switch (key_words * 100 + cw_build * 10 + cw_probe) {
    case 100: join_bcast_t<1, 0, 0>(L, bp, grid); break;
    case 101: join_bcast_t<1, 0, 1>(L, bp, grid); break;
    case 102: join_bcast_t<1, 0, 2>(L, bp, grid); break;
    case 110: join_bcast_t<1, 1, 0>(L, bp, grid); break;
    case 111: join_bcast_t<1, 1, 1>(L, bp, grid); break;
    case 112: join_bcast_t<1, 1, 2>(L, bp, grid); break;
    case 120: join_bcast_t<1, 2, 0>(L, bp, grid); break;
    case 121: join_bcast_t<1, 2, 1>(L, bp, grid); break;
    case 122: join_bcast_t<1, 2, 2>(L, bp, grid); break;
    case 103: join_bcast_t<1, 0, 3>(L, bp, grid); break;
    case 113: join_bcast_t<1, 1, 3>(L, bp, grid); break;
    case 123: join_bcast_t<1, 2, 3>(L, bp, grid); break;
    case 130: join_bcast_t<1, 3, 0>(L, bp, grid); break;
    case 131: join_bcast_t<1, 3, 1>(L, bp, grid); break;
    case 132: join_bcast_t<1, 3, 2>(L, bp, grid); break;
    case 133: join_bcast_t<1, 3, 3>(L, bp, grid); break;
    case 200: join_bcast_t<2, 0, 0>(L, bp, grid); break;
    case 201: join_bcast_t<2, 0, 1>(L, bp, grid); break;
    case 202: join_bcast_t<2, 0, 2>(L, bp, grid); break;
    case 210: join_bcast_t<2, 1, 0>(L, bp, grid); break;
    case 211: join_bcast_t<2, 1, 1>(L, bp, grid); break;
    case 212: join_bcast_t<2, 1, 2>(L, bp, grid); break;
    case 220: join_bcast_t<2, 2, 0>(L, bp, grid); break;
    case 221: join_bcast_t<2, 2, 1>(L, bp, grid); break;
    case 222: join_bcast_t<2, 2, 2>(L, bp, grid); break;
    default: launch_failed("join_broadcast", "no kernel for this key/carry word count", true);
    }
}

void launch_debug_stall(const Launch& L, uint32_t ms) {
    int dev = 0, khz = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
    RJ_KLAUNCH(L, "debug_stall", k_debug_stall, 1, 64, (unsigned long long)ms * (unsigned long long)khz);
}

void launch_pack_validity(const Launch& L, const uint8_t* v0, const uint8_t* v1, const uint8_t* v2, uint32_t n,
                          uint32_t* out) {
    if (!n) return;
    RJ_KLAUNCH(L, "pack_validity", k_pack_validity, (n + 255) / 256, 256, v0, v1, v2, n, out);
}

void launch_split_records(const Launch& L, const SplitParams& sp, uint64_t n) {
    if (!n) return;
    RJ_KLAUNCH(L, "split_records", k_split_records, (uint32_t)((n + 255) / 256), 256, sp, n);
}

void launch_gather(const Launch& L, const ColRef& src, const uint32_t* idx, uint64_t n,
                   const OutStream& dst, uint8_t* dst_valid) {
    if (!n) return;
    uint32_t grid = (uint32_t)((n + 255) / 256);
    if (src.width == 4)
        RJ_KLAUNCH(L, "gather", (k_gather<4>), grid, 256, src, idx, n, dst, dst_valid);
    else
        RJ_KLAUNCH(L, "gather", (k_gather<8>), grid, 256, src, idx, n, dst, dst_valid);
}

void launch_finish_pages(const Launch& L, uint8_t* pages, uint64_t n_rows, int width) {
    if (!n_rows) return;
    uint32_t rf = width == 4 ? ROWS32 : ROWS64;
    uint32_t np = (uint32_t)((n_rows + rf - 1) / rf);
    RJ_KLAUNCH(L, "finish_pages", k_finish_pages, np, 256, pages, n_rows, rf);
}

void launch_finish_streams(const Launch& L, uint8_t* const* pages, const int* widths, uint32_t n,
                           const unsigned long long* n_rows_dev, uint64_t cap_rows) {
    if (!n || !cap_rows) return;
    FinishStreams fs{};
    fs.n = n;
    uint64_t max_pages = 0;
    for (uint32_t i = 0; i < n && i < 3; ++i) {
        fs.pages[i] = pages[i];
        fs.rows_full[i] = widths[i] == 4 ? ROWS32 : ROWS64;
        max_pages = std::max<uint64_t>(max_pages, (cap_rows + fs.rows_full[i] - 1) / fs.rows_full[i]);
    }
    RJ_KLAUNCH(L, "finish_pages", k_finish_streams, dim3((uint32_t)((max_pages + 3) / 4), n), 256, fs,
               n_rows_dev, cap_rows);
}

void launch_encode_nullable(const Launch& L, const uint8_t* values, const uint8_t* valid,
                            uint64_t n_rows, int width, uint8_t* pages) {
    if (!n_rows) return;
    uint32_t rf = width == 4 ? ROWS32 : ROWS64;
    uint32_t np = (uint32_t)((n_rows + rf - 1) / rf);
    if (width == 4)
        RJ_KLAUNCH(L, "encode_nullable", (k_encode_nullable<4>), np, 256, values, valid, n_rows,
                   pages);
    else
        RJ_KLAUNCH(L, "encode_nullable", (k_encode_nullable<8>), np, 256, values, valid, n_rows,
                   pages);
}

}  // namespace rj
