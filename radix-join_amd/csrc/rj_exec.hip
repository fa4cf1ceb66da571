// rj_exec.hip — plan executor: the host logic of Contest::execute on MI355X.
//
// Replaces reference src/execute.cpp:266-324 (execute, execute_impl, execute_scan,
// execute_hash_join, hash_join_omp).  Shape: columnar + late materialisation.
//   * A node's result ("Rel") is a set of device columns: regular page images
//     addressed in place, dense arrays, or row-id columns standing for VARCHAR.
//   * A JoinNode radix-partitions (hashed key, carry) tuples of both children
//     with the same bit plan, joins co-partitions in LDS and emits up to three
//     streams: key, build carry, probe carry.  The carry is the single payload
//     column itself when a side needs only one (no gather afterwards), else the
//     row index into the child, gathered per output column.
//   * At the root the streams are written straight into Page images.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <functional>
#include <set>

#include "rj_internal.hpp"
#include "rj_xplan.hpp"

namespace rj {

namespace {

struct DCol {
    int32_t        type = RJ_INT32;  // declared DataType
    int32_t        kind = COL_NONE;
    int32_t        width = 4;  // bytes per value on device (VARCHAR: 4 = row id)
    const uint8_t* ptr = nullptr;
    const uint8_t* valid = nullptr;
    BufP           hold, hold_valid;
    int            vc_table = -1, vc_col = -1;  // VARCHAR provenance (base table, column)
    const TableColumn* tcol = nullptr;          // scan passthrough
    ColRef ref() const { return ColRef{ptr, valid, kind, width}; }
};

struct Rel {
    uint64_t          n = 0;
    std::vector<DCol> cols;
};

struct Parted {
    int      NW = 0;
    BufP     wbuf[MAX_WORDS];
    Words    w{};
    BufP     off;  // u32[NP+1]
    uint32_t NP = 0;
    std::vector<uint32_t> pbits;  // radix bits of each pass
    bool     packed = false;      // w[0] = {hashed key, carry} pairs, no w[1]
    bool     aos3 = false;        // w[0] = 12-byte {hashed key, carry lo, carry hi} tuples, no w[1..2]
    uint64_t n_tuples = 0;        // tuples that went in (NULL keys are dropped on the way: an upper bound)
};

// Tuples that already sit in the partition layout (hashed word arrays, a pair array for a
// two-word carry, or one array of packed {key, carry} pairs): what a rank holds after the
// exchange step of a sharded join.
struct WordSrc {
    Words    w{};
    uint64_t n = 0;
    bool     packed = false;
};

// How the FIRST pass of a partition() call differs from the plain "one segment, one bit field"
// case — both forms belong to the sharded join:
//   * composite digit (stage A): owner rank from the top hash bits, first local digit from the
//     low ones, in one pass (PassParams::hi_shift / lo_bits);
//   * pre-segmented input (stage B): what arrived in the exchange is a set of runs, one per
//     (source rank, local digit), listed digit-major; `world` of them feed one output segment
//     (ExchangePlan::seg_begin / seg_end / part_off, rj_xplan.hpp).
struct PassShape {
    uint32_t        hi_shift = 0, lo_bits = 0;
    const uint32_t* seg_begin = nullptr;  // device, [nseg]
    const uint32_t* seg_end = nullptr;    // device, [nseg]
    uint32_t        nseg = 0;
    uint32_t        oseg_shift = 0;       // input segment i feeds output segment i >> oseg_shift
    const uint32_t* oseg_off = nullptr;   // device, [n_oseg + 1]
    uint32_t        n_oseg = 0;
    std::vector<uint32_t> prior_bits;     // digits consumed before this call (for the partition index)
};

struct JoinSpec {
    bool                  build_left = true;
    uint64_t              left_attr = 0, right_attr = 0;
    std::vector<uint64_t> out_idx;
    std::vector<int32_t>  out_type;
    bool                  prehashed = false;
    int                   forced_bits = 0;
    uint32_t              top_bits_taken = 0;  // high hash bits that are constant in this input
};

uint32_t ceil_log2(uint64_t v) {
    uint32_t b = 0;
    while ((uint64_t(1) << b) < v) ++b;
    return b;
}

class Exec {
   public:
    Exec(Context* c, const rj_plan* p, Table* const* t, uint64_t nt, int fl)
        : ctx(c), plan(p), tables(t), n_tables(nt), flags(fl), L(c->launch()) {}
    void set_fetch(TableFetch* f) { fetch = f; }

    Result* run() {
        if (!plan || plan->root >= plan->n_nodes) throw_fmt(RJ_ERR_ARG, "bad plan root");
        std::unique_ptr<Result> res(new rj_result());
        res->ctx = ctx;
        const rj_node& root = plan->nodes[plan->root];
        if (root.kind == RJ_NODE_SCAN) {
            root_scan(root, *res);
        } else {
            (void)node(plan->root, res.get(), 0);
        }
        ctx->sync();
        return res.release();
    }

    // rj_join_tuples: same join core over caller-provided dense tuples
    Result* run_tuples(const rj_tuples* b, const rj_tuples* p, uint32_t skip_rank_bits) {
        std::unique_ptr<Result> res(new rj_result());
        res->ctx = ctx;
        auto mk = [](const rj_tuples* t) {
            Rel r;
            r.n = t->n;
            DCol k;
            k.type = RJ_INT32;
            k.kind = COL_DENSE;
            k.width = 4;
            k.ptr = static_cast<const uint8_t*>(t->key);
            DCol c = k;
            c.ptr = static_cast<const uint8_t*>(t->carry);
            r.cols = {k, c};
            return r;
        };
        if (b->n > 0xfffffff0ull || p->n > 0xfffffff0ull)
            throw_fmt(RJ_ERR_UNSUPPORTED, "more than 2^32 tuples");
        if (b->hashed != p->hashed) throw_fmt(RJ_ERR_ARG, "build/probe disagree on `hashed`");
        Rel      lb = mk(b), rp = mk(p);
        JoinSpec js;
        js.build_left = true;
        js.out_idx = {0, 1, 3};
        js.out_type = {RJ_INT32, RJ_INT32, RJ_INT32};
        js.prehashed = b->hashed != 0;
        // the top hash bits stage A consumed are constant on this rank: the radix plan stays
        // below them (they would only produce empty partitions)
        if (skip_rank_bits > 16) throw_fmt(RJ_ERR_ARG, "skip_rank_bits > 16");
        js.top_bits_taken = js.prehashed ? skip_rank_bits : 0u;
        (void)join_core(lb, rp, js, res.get());
        ctx->sync();
        return res.release();
    }

    // rj_shard_partition: decode + hash + one pass over the TOP log2(n_ranks) hash bits
    void run_shard(const Table* t, uint64_t key_col, uint64_t carry_col, uint32_t n_ranks,
                   rj_tuples* out, uint64_t* counts) {
        if (n_ranks == 0 || (n_ranks & (n_ranks - 1)) || n_ranks > (1u << PT_MAXBITS))
            throw_fmt(RJ_ERR_UNSUPPORTED, "n_ranks must be a power of two <= %d", 1 << PT_MAXBITS);
        if (key_col >= t->cols.size() || carry_col >= t->cols.size())
            throw_fmt(RJ_ERR_ARG, "column out of range");
        DCol k = table_col(t, (int)key_col), c = table_col(t, (int)carry_col);
        if (k.type != RJ_INT32 || c.type != RJ_INT32)
            throw_fmt(RJ_ERR_UNSUPPORTED, "sharded path carries INT32 key + INT32 payload");
        if (c.valid) throw_fmt(RJ_ERR_UNSUPPORTED, "sharded path: payload column has NULLs");
        TupleSrc src{};
        src.key = k.ref();
        src.carry = c.ref();
        src.n_rows = (uint32_t)t->num_rows;
        src.carry_mode = CARRY_COLUMN;
        if (!out->key || !out->carry) throw_fmt(RJ_ERR_ARG, "rj_shard_partition: null output buffers");
        uint32_t rb = ceil_log2(n_ranks);
        Words    ext{};
        ext.w[0] = static_cast<uint32_t*>(out->key);
        ext.w[1] = static_cast<uint32_t*>(out->carry);
        Parted P = partition(&src, nullptr, 1, 1, rb, /*shift0=*/rb ? 32 - rb : 31, /*single pass*/ true, &ext);
        std::vector<uint32_t> off(n_ranks + 1);
        RJ_HIP(hipMemcpyAsync(off.data(), P.off->p, (n_ranks + 1) * 4, hipMemcpyDeviceToHost,
                              ctx->stream));
        ctx->sync();
        for (uint32_t r = 0; r < n_ranks; ++r) counts[r] = off[r + 1] - off[r];
        out->n = off[n_ranks];
        out->hashed = 1;
        out->reserved = 0;
    }

    // ---- state and building blocks (a ShardedExec drives one Exec per rank through them)
    Context*       ctx;
    const rj_plan* plan;
    Table* const*  tables;
    uint64_t       n_tables;
    int            flags;
    TableFetch*    fetch = nullptr;
    Launch         L;
    std::map<std::pair<const Table*, int>, DCol> decoded_;
    std::map<std::pair<int, int>, std::vector<uint64_t>> vc_dir_;  // VARCHAR page directories

    // ------------------------------------------------------------- scan side
    DCol table_col(const Table* t, int c) {
        const TableColumn& tc = t->cols[c];
        if (tc.skipped) throw_fmt(RJ_ERR_ARG, "column was not uploaded (not referenced by a scan)");
        DCol               d;
        d.type = tc.type;
        d.tcol = &tc;
        if (tc.type == RJ_VARCHAR) {
            d.kind = COL_IOTA;
            d.width = 4;
            return d;
        }
        d.width = tc.type == RJ_INT32 ? 4 : 8;
        if (t->num_rows == 0) {  // nothing to address (an empty shard): no validity to carry either
            d.kind = COL_DENSE;
            return d;
        }
        if (tc.regular) {
            d.kind = COL_PAGED;
            d.ptr = tc.dev_pages;
            return d;
        }
        auto key = std::make_pair(t, c);
        auto it = decoded_.find(key);
        if (it != decoded_.end()) return it->second;
        // K1: irregular pages (NULLs, short pages) -> dense values + validity
        uint64_t n = t->num_rows;
        d.kind = COL_DENSE;
        d.hold = ctx->buf(n * d.width);
        d.hold_valid = ctx->buf(n);
        RJ_HIP(hipMemsetAsync(d.hold->p, 0, n * d.width, ctx->stream));
        RJ_HIP(hipMemsetAsync(d.hold_valid->p, 0, n, ctx->stream));
        if (tc.n_pages) {
            BufP row_base = ctx->buf((tc.n_pages + 1) * 4);
            launch_scan_bins(L, tc.page_rows->as<uint32_t>(), (uint32_t)tc.n_pages,
                             row_base->as<uint32_t>(), nullptr);
            launch_decode_pages(L, tc.dev_pages, (uint32_t)tc.n_pages, d.width,
                                row_base->as<uint32_t>(), n, d.hold->as<uint8_t>(),
                                d.hold_valid->as<uint8_t>());
        }
        d.ptr = d.hold->as<uint8_t>();
        d.valid = d.hold_valid->as<uint8_t>();
        decoded_[key] = d;
        return d;
    }

    const Table* table_of(const rj_node& n) {
        return table_by_id(n.base_table_id);
    }
    const Table* table_by_id(uint64_t id) {
        if (id >= n_tables) throw_fmt(RJ_ERR_ARG, "scan: bad base_table_id");
        return fetch ? fetch->get(id) : tables[id];
    }

    // execute_scan (reference src/execute.cpp:284-300): column selection, zero copy
    Rel scan(const rj_node& n) {
        const Table* t = table_of(n);
        Rel          r;
        r.n = t->num_rows;
        for (uint64_t k = 0; k < n.n_out; ++k) {
            uint64_t c = n.out_idx[k];
            if (c >= t->cols.size()) throw_fmt(RJ_ERR_ARG, "scan: output attr out of range");
            if (t->cols[c].type != n.out_type[k])
                throw_fmt(RJ_ERR_ARG, "scan: declared type differs from the column's type");
            DCol d = table_col(t, (int)c);
            if (d.type == RJ_VARCHAR) {
                d.vc_table = (int)n.base_table_id;
                d.vc_col = (int)c;
            }
            r.cols.push_back(d);
        }
        return r;
    }

    // A plan whose root is a Scan: the result pages are the input pages.
    void root_scan(const rj_node& n, Result& res) {
        const Table* t = table_of(n);
        res.num_rows = t->num_rows;
        for (uint64_t k = 0; k < n.n_out; ++k) {
            uint64_t c = n.out_idx[k];
            if (c >= t->cols.size()) throw_fmt(RJ_ERR_ARG, "scan: output attr out of range");
            const TableColumn& tc = t->cols[c];
            if (tc.type != n.out_type[k])
                throw_fmt(RJ_ERR_ARG, "scan: declared type differs from the column's type");
            ResultColumn rc;
            rc.type = tc.type;
            rc.n_pages = tc.n_pages;
            if (tc.skipped) throw_fmt(RJ_ERR_ARG, "column was not uploaded");
            if (tc.type == RJ_VARCHAR) {
                rc.n_pages = tc.vc_pages.size();
                rc.host_pages.resize(rc.n_pages * PAGE_BYTES);
                for (uint64_t pg = 0; pg < rc.n_pages; ++pg)
                    memcpy(rc.host_pages.data() + pg * PAGE_BYTES, tc.vc_pages[pg], PAGE_BYTES);
            } else if (tc.n_pages) {
                rc.dev_pages = ctx->buf(tc.n_pages * PAGE_BYTES);
                RJ_HIP(hipMemcpyAsync(rc.dev_pages->p, tc.dev_pages, tc.n_pages * PAGE_BYTES,
                                      hipMemcpyDeviceToDevice, ctx->stream));
            }
            res.cols.push_back(std::move(rc));
        }
    }

    // ------------------------------------------------------------- plan walk
    // execute_impl (reference src/execute.cpp:302-314); children left first (:48-49)
    Rel node(uint64_t idx, Result* root_res, int depth) {
        if (idx >= plan->n_nodes) throw_fmt(RJ_ERR_ARG, "bad node index");
        if (depth > 4096) throw_fmt(RJ_ERR_ARG, "plan too deep (cycle?)");
        const rj_node& n = plan->nodes[idx];
        if (n.kind == RJ_NODE_SCAN) return scan(n);
        if (n.kind != RJ_NODE_JOIN) throw_fmt(RJ_ERR_ARG, "bad node kind");
        Rel      l = node(n.left, nullptr, depth + 1);
        Rel      r = node(n.right, nullptr, depth + 1);
        JoinSpec js;
        js.build_left = n.build_left != 0;
        js.left_attr = n.left_attr;
        js.right_attr = n.right_attr;
        js.out_idx.assign(n.out_idx, n.out_idx + n.n_out);
        js.out_type.assign(n.out_type, n.out_type + n.n_out);
        js.forced_bits = ctx->radix_bits_override;
        return join_core(l, r, js, root_res);
    }

    // -------------------------------------------------------- radix partition
    static uint32_t tiles_per_group(uint64_t tuples_per_segment, uint64_t target_groups) {
        uint64_t k = (tuples_per_segment + (uint64_t)PT_TILE * target_groups - 1) /
                     ((uint64_t)PT_TILE * target_groups);
        return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(k, 1), 8);
    }

    // Source = columns (`src`: page decode, NULL drop and hashing ride on the first pass) or
    // tuples already in the partition layout (`ws`, sharded stage B).
    // `external`: caller-owned arrays that receive the output of a single-pass partition
    // after_offsets (optional): called once the FIRST pass' partition offsets are enqueued to be
    // computed, before its scatter is enqueued — a caller that needs the counts on the host
    // (the sharded join's stage A) copies them out ahead of the scatter.
    Parted partition(const TupleSrc* srcp, const WordSrc* ws, int KW, int CW, uint32_t bits,
                     uint32_t shift0 = 0, bool single_pass = false, const Words* external = nullptr,
                     const std::function<void(const uint32_t*)>* after_offsets = nullptr,
                     const PassShape* shape = nullptr) {
        Parted P;
        const bool preseg = shape && shape->nseg > 0;
        P.NW = KW + CW;
        const TupleSrc none{};
        const TupleSrc& src = srcp ? *srcp : none;
        const uint64_t  n = ws ? ws->n : src.n_rows;
        uint32_t       passes = single_pass ? 1 : (bits + PT_MAXBITS - 1) / PT_MAXBITS;
        if (passes == 0) passes = 1;
        std::vector<uint32_t> pbits(passes, bits / passes);
        for (uint32_t i = 0; i < bits % passes; ++i) pbits[i]++;
        // tuning knobs (experiments): RJ_TUNE_P1_BITS moves bits between pass 1 and pass 2
        if (passes == 2 && ctx->tune.p1_bits > 0) {
            uint32_t b1 = (uint32_t)ctx->tune.p1_bits;
            if (b1 < bits && b1 <= PT_CAPBITS && bits - b1 <= PT_CAPBITS) {
                pbits[0] = b1;
                pbits[1] = bits - b1;
            }
        }

        // Two passes whose final partitions fit one LDS histogram: both digits are counted in a
        // single read of the source, the scatters reserve their ranges tile by tile, and the
        // second histogram pass (4 B/tuple) disappears.
        const bool fine = passes == 2 && bits <= (uint32_t)PT_FINEBITS && !external && !shape &&
                          ctx->tune.fine != 0;
        // key + one carry word travel as 8-byte pairs in one array (half the streams, twice the
        // bytes per run); RJ_TUNE_PACK: 0 = never, 1 = every plan, 2 = fine-histogram plans only
        // (a later pass' plain histogram reads 8 instead of 4 bytes per tuple from pairs)
        const int pack_mode = ctx->tune.pack;
        P.packed = ws ? ws->packed
                      : (!external && KW == 1 && CW == 1 && (pack_mode == 1 || (pack_mode == 2 && fine)));
        // one key word + a two-word carry: the LAST pass writes 12-byte tuples into one array
        // (RJ_TUNE_AOS3=0: key array + pair array, as the earlier passes do)
        P.aos3 = KW == 1 && CW == 2 && !external && !single_pass && ctx->tune.aos3 != 0;
        // ... and so do the passes before it (one output stream of 384-byte runs instead of a
        // 128-byte-run key stream + a 256-byte-run pair stream; a later pass' histogram then
        // reads a 16-bit digit side array instead of the keys)
        // RJ_TUNE_AOS_MID: 0 never, 1 always, 2 (default) only for fine-histogram plans, which
        // need no digit side array — with it the first pass loses more (+0.8 ms at 1 B rows) than
        // the histogram gains (−0.4 ms): profiles/r02_y_aos_between_passes_ab.log
        // RJ_TUNE_AOS_MID=3: always, and WITHOUT the side array (the next histogram reads the keys
        // out of the 12-byte tuples)
        const bool aos_mid = P.aos3 && passes >= 2 &&
                             (ctx->tune.aos_mid == 1 || ctx->tune.aos_mid == 3 || (ctx->tune.aos_mid == 2 && fine));
        const bool mid_side = aos_mid && !fine && ctx->tune.aos_mid != 3;
        BufP  A[MAX_WORDS], B[MAX_WORDS], AOS, MID[2], SIDE[2];
        Words wa{}, wb{}, waos{};
        if (P.aos3) {
            AOS = ctx->buf(std::max<uint64_t>(n, 1) * 12);
            waos.w[0] = AOS->as<uint32_t>();
        }
        if (aos_mid) {
            for (uint32_t k = 0; k < std::min<uint32_t>(passes - 1, 2); ++k) {
                MID[k] = ctx->buf(std::max<uint64_t>(n, 1) * 12);
                if (mid_side) SIDE[k] = ctx->buf(std::max<uint64_t>(n, 1) * 2 + 16);
            }
        }
        // packed pairs above the fine histogram's limit: a later pass' histogram would read the
        // 8-byte pairs for their 4-byte keys — the pass before it writes the NEXT digit of every
        // tuple as a 16-bit side array instead (2 bytes written + 2 read per tuple instead of 8 read)
        const bool packed_side = P.packed && !fine && passes >= 2 && ctx->tune.packed_side != 0 &&
                                 n >= (uint64_t)ctx->tune.xcd_min_rows;
        if (packed_side)
            for (uint32_t k = 0; k < std::min<uint32_t>(passes - 1, 2); ++k)
                SIDE[k] = ctx->buf(std::max<uint64_t>(n, 1) * 2 + 16);
        // the last two words of a carry of two or three words are ONE array of 8-byte pairs (at word
        // pair_word); key + one carry word one array of pairs altogether (packed)
        const int pair_word = CW >= 2 ? KW + CW - 2 : -1;
        for (int a = 0; a < (P.packed ? 1 : P.NW); ++a) {
            if (pair_word >= 0 && a == pair_word + 1) continue;
            const uint64_t wbytes = (P.packed || a == pair_word) ? 8 : 4;
            if (external) {
                wa.w[a] = external->w[a];
                continue;
            }
            if ((P.aos3 && passes == 1) || aos_mid) continue;  // every pass writes 12-byte tuples
            // (+2048: a blocked array ends with a whole block, see BlockedLoader)
            A[a] = ctx->buf(n * wbytes + 2048);
            wa.w[a] = A[a]->as<uint32_t>();
            if (passes > (P.aos3 ? 2u : 1u)) {
                B[a] = ctx->buf(n * wbytes + 2048);
                wb.w[a] = B[a]->as<uint32_t>();
            }
        }
        BufP     seg_off;  // offsets produced by the previous pass
        uint32_t nseg = 1, shift = shift0;
        Words    cur = ws ? ws->w : Words{}, nxt = (P.aos3 && passes == 1) ? waos : wa;
        bool     cur_is_a = false;
        BufP       fine_off, fine_cursor, coarse_off, coarse_cursor;
        // tiles per group of the first pass (the fine histogram must know it: see fine_xcd)
        uint32_t tpg_first = tiles_per_group(n, 4096);
        if (ctx->tune.tpg1 > 0) tpg_first = (uint32_t)ctx->tune.tpg1;
        bool fine_xcd = false;
        if (fine) {
            const uint32_t NB = 1u << bits, F1 = 1u << pbits[0], F2 = 1u << pbits[1];
            BufP           fh = ctx->buf((uint64_t)NB * 4);
            const uint64_t tiles = (n + PT_TILE - 1) / PT_TILE;
            uint32_t       fgrid = (uint32_t)std::min<uint64_t>(tiles, (uint64_t)ctx->compute_units());
            // XCD-aware placement of the first pass: per-sub-range counts out of the histogram
            // (its grid then is a multiple of 8: workgroup w counts for sub-range w & 7)
            fine_xcd = ctx->tune.xcd_split && n >= (uint64_t)ctx->tune.xcd_min_rows && fgrid >= 8;
            if (fine_xcd) fgrid &= ~7u;
            fine_off = ctx->buf(((uint64_t)NB + 1) * 4);
            fine_cursor = ctx->buf((uint64_t)NB * 4);
            coarse_off = ctx->buf(((uint64_t)F1 + 1) * 4);
            coarse_cursor = ctx->buf((uint64_t)F1 * 4 * (fine_xcd ? 8 : 1));
            BufP coarse_x;
            RJ_HIP(hipMemsetAsync(fh->p, 0, (uint64_t)NB * 4, ctx->stream));
            if (fine_xcd) {
                coarse_x = ctx->buf((uint64_t)F1 * 8 * 4);
                RJ_HIP(hipMemsetAsync(coarse_x->p, 0, (uint64_t)F1 * 8 * 4, ctx->stream));
            }
            uint32_t* cx = fine_xcd ? coarse_x->as<uint32_t>() : nullptr;
            if (ws)
                launch_fine_hist_words(L, ws->w, ws->packed, (uint32_t)n, shift0, pbits[0], pbits[1], fgrid,
                                       fh->as<uint32_t>(), fine_xcd ? tpg_first : 0u, cx);
            else
                launch_fine_hist_src(L, src, KW, shift0, pbits[0], pbits[1], fgrid, fh->as<uint32_t>(),
                                     fine_xcd ? tpg_first : 0u, cx);
            launch_scan_fine(L, fh->as<uint32_t>(), F1, F2, fine_off->as<uint32_t>(),
                             fine_cursor->as<uint32_t>(), coarse_off->as<uint32_t>(),
                             coarse_cursor->as<uint32_t>(), cx);
        }
        // RJ_TUNE_MALL_CHUNK: the second pass of a big packed two-pass plan runs chunk by chunk (below)
        struct Ev2 {
            hipEvent_t e = nullptr;
            void       make() { RJ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
            ~Ev2() {
                if (e) (void)hipEventDestroy(e);
            }
        } chunk_ev;
        const bool chunk_second = ctx->tune.mall_chunk > 0 && passes == 2 && !fine && !shape && !ws && !external && P.packed &&
                                  !packed_side && ctx->tune.xcd_split && n >= (uint64_t)ctx->tune.xcd_min_rows;
        // RJ_TUNE_BLOCKED_MID: between the passes of a packed plan the pairs lie in blocks of 16 keys + 16 carries
        // (BlockedLoader), so that the next pass' histogram reads 4 instead of 8 bytes per tuple
        const bool blocked_mid = ctx->tune.blocked_mid != 0 && P.packed && passes >= 2 && !fine && !shape && !ws && !external &&
                                 !packed_side && !chunk_second;
        for (uint32_t p = 0; p < passes; ++p) {
            const uint32_t F = 1u << pbits[p];
            PassParams     pp{};
            pp.nseg = nseg;
            pp.n = (uint32_t)n;
            pp.shift = shift;
            pp.fanout_log2 = pbits[p];
            uint32_t n_groups;
            BufP     grp_start;
            // segments this pass reads / output segments it fills (the same unless the input is a
            // set of exchanged runs, several of which feed one output segment)
            const bool      first_preseg = p == 0 && preseg;
            const uint32_t  nseg_in = first_preseg ? shape->nseg : nseg;
            const uint32_t  n_oseg = first_preseg ? shape->n_oseg : nseg;
            const uint64_t  bins = (uint64_t)n_oseg * F;
            if (p == 0 && shape && shape->hi_shift) {
                pp.hi_shift = shape->hi_shift;
                pp.lo_bits = shape->lo_bits;
            }
            if (p == 0 && !preseg) {
                pp.tiles_per_group = tpg_first;
                uint64_t gt = (uint64_t)pp.tiles_per_group * PT_TILE;
                n_groups = (uint32_t)((n + gt - 1) / gt);
            } else {
                    pp.tiles_per_group = tiles_per_group(n / nseg_in + 1, 16);
                    // the last pass of 12-byte-tuple plans (one 384-byte-run output stream) does
                    // better with short groups — 2 tiles per workgroup: 9.3-10.3 -> 8.9-9.1 ms per step
                    // at 1 B rows; packed 8-byte tuples lose with them
                    // (profiles/r03_l_later_pass_group_size_ab.log)
                    if (P.aos3 && p + 1 == passes) pp.tiles_per_group = std::min<uint32_t>(pp.tiles_per_group, 2u);
                    uint64_t gt = (uint64_t)pp.tiles_per_group * PT_TILE;
                    if (ctx->tune.tpg2 > 0) {
                        pp.tiles_per_group = (uint32_t)ctx->tune.tpg2;
                        gt = (uint64_t)pp.tiles_per_group * PT_TILE;
                    }
                    n_groups = (uint32_t)(n / gt + nseg_in);  // upper bound; exact count lives on device
                    if (ctx->tune.xcd_split && n >= (uint64_t)ctx->tune.xcd_min_rows) {
                        pp.xcd_remap = 1;
                        n_groups += 8;  // 8 * ceil(G / 8) workgroups
                    }
                    const uint32_t* in_off = first_preseg ? shape->seg_begin : seg_off->as<uint32_t>();
                    grp_start = ctx->buf(((uint64_t)nseg_in + 1) * 4);
                    launch_group_table(L, in_off, nseg_in, (uint32_t)gt, grp_start->as<uint32_t>(),
                                       first_preseg ? shape->seg_end : nullptr);
                    pp.nseg = nseg_in;
                    pp.seg_off = in_off;
                    pp.seg_end = first_preseg ? shape->seg_end : nullptr;
                    pp.oseg_shift = first_preseg ? shape->oseg_shift : 0u;
                    pp.grp_start = grp_start->as<uint32_t>();
                }
                BufP off, hist, cursor;  // this pass' partition offsets, bin totals, write cursors
                if (fine) {  // offsets and cursors of both passes came out of the fine histogram
                    off = p == 0 ? coarse_off : fine_off;
                    cursor = p == 0 ? coarse_cursor : fine_cursor;
                    pp.cursor = cursor->as<uint32_t>();
                    pp.xcd_log2 = (p == 0 && fine_xcd) ? 3u : 0u;
                } else {
                    // sub-ranges per XCD only where a partition gets many runs (big inputs)
                    pp.xcd_log2 = (ctx->tune.xcd_split && p == 0 && !preseg && n >= (uint64_t)ctx->tune.xcd_min_rows) ? 3u : 0u;
                    hist = ctx->buf((bins << pp.xcd_log2) * 4);
                    off = ctx->buf((bins + 1) * 4);
                    cursor = ctx->buf((bins << pp.xcd_log2) * 4);
                    RJ_HIP(hipMemsetAsync(hist->p, 0, (bins << pp.xcd_log2) * 4, ctx->stream));
                    pp.hist = hist->as<uint32_t>();
                    pp.cursor = cursor->as<uint32_t>();
                    if (p == 1 && chunk_second) {
                        // ---- histogram / scan / scatter of this pass, `mall_chunk` input segments at a time,
                        // chunks alternating between the context's two compute streams: the scatter of a
                        // chunk re-reads what its histogram launch has just pulled through the Infinity
                        // Cache.  Segments never share bins, cursors or output ranges, so the chunks are
                        // independent; grids are exact (the segment sizes are on the host by now).
                        RJ_HIP(hipEventSynchronize(chunk_ev.e));
                        const uint32_t* hoff = static_cast<const uint32_t*>(ctx->small_pinned());
                        const uint64_t  gt = (uint64_t)pp.tiles_per_group * PT_TILE;
                        std::vector<uint32_t> gstart(nseg_in + 1, 0);
                        for (uint32_t sgi = 0; sgi < nseg_in; ++sgi)
                            gstart[sgi + 1] = gstart[sgi] + (uint32_t)(((uint64_t)(hoff[sgi + 1] - hoff[sgi]) + gt - 1) / gt);
                        Ev2 e_in, e_aux;
                        e_in.make();
                        e_aux.make();
                        hipStream_t aux = ctx->aux_stream();
                        RJ_HIP(hipEventRecord(e_in.e, ctx->stream));  // (behind the previous scatter and the bin memset)
                        RJ_HIP(hipStreamWaitEvent(aux, e_in.e, 0));
                        Launch L2 = L;
                        L2.stream = aux;
                        const uint32_t CH = (uint32_t)ctx->tune.mall_chunk;
                        uint32_t       c = 0;
                        for (uint32_t s0 = 0; s0 < nseg_in; s0 += CH, ++c) {
                            const uint32_t s1 = std::min(nseg_in, s0 + CH), G = gstart[s1] - gstart[s0];
                            const Launch& Lc = (c & 1u) ? L2 : L;
                            PassParams    pc = pp;
                            pc.nseg = s1 - s0;
                            pc.seg_off = pp.seg_off + s0;
                            pc.grp_start = pp.grp_start + s0;
                            pc.grp_base = gstart[s0];
                            pc.hist = pp.hist + (size_t)s0 * F;
                            pc.cursor = pp.cursor + (size_t)s0 * F;
                            const uint32_t grid = pp.xcd_remap ? ((G + 7u) & ~7u) : G;
                            if (G) launch_pass_hist_packed(Lc, cur.w[0], pc, grid);
                            // (empty segments still need their partition offsets)
                            launch_scan_segments(Lc, pc.hist, pc.seg_off, pc.nseg, F, 0, off->as<uint32_t>() + (size_t)s0 * F, pc.cursor);
                            if (G) launch_pass_scatter_packed(Lc, cur.w[0], pc, grid, nxt.w[0]);
                        }
                        RJ_HIP(hipEventRecord(e_aux.e, aux));
                        RJ_HIP(hipStreamWaitEvent(ctx->stream, e_aux.e, 0));  // (before any buffer of this pass is reused)
                        seg_off = off;
                        nseg = (uint32_t)bins;
                        shift += pbits[p];
                        cur = nxt;
                        cur_is_a = (p % 2 == 0);
                        nxt = cur_is_a ? wb : wa;
                        continue;
                    }
                    if (p == 0 && !ws)
                        launch_pass_hist_src(L, src, KW, pp, n_groups);
                    else if ((mid_side || packed_side) && p > 0)
                        launch_pass_hist_digits(L, SIDE[(p - 1) % 2]->as<uint16_t>(), pp, n_groups);
                    else if (aos_mid && p > 0)
                        launch_pass_hist_aos3(L, MID[(p - 1) % 2]->as<uint32_t>(), pp, n_groups);
                    else if (P.packed && blocked_mid && p > 0)
                        launch_pass_hist_blocked(L, cur.w[0], pp, n_groups);
                    else if (P.packed)
                        launch_pass_hist_packed(L, cur.w[0], pp, n_groups);
                    else
                        launch_pass_hist_dense(L, cur, pp, n_groups);
                    // (one workgroup per OUTPUT segment; its base = where that segment starts)
                    launch_scan_segments(L, pp.hist, first_preseg ? shape->oseg_off : (p == 0 ? nullptr : pp.seg_off),
                                         n_oseg, F, pp.xcd_log2, off->as<uint32_t>(), pp.cursor);
                }
                if (p == 0 && after_offsets) (*after_offsets)(off->as<uint32_t>());
                // the second pass in chunks needs this pass' partition sizes on the HOST (exact grids per
                // chunk): they leave now, before the scatter is enqueued, and are read while it runs
                if (p == 0 && chunk_second) {
                    if ((bins + 1) * 4 > Context::SMALL_PINNED / 4) throw_fmt(RJ_ERR_DEVICE, "chunked pass: too many segments");
                    RJ_HIP(hipMemcpyAsync(ctx->small_pinned(), off->p, (bins + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
                    chunk_ev.make();
                    RJ_HIP(hipEventRecord(chunk_ev.e, ctx->stream));
                }
                BufP pt_diag;
                if (ctx->tune.diag >= 3) {  // phase stamps of the scatter (diagnostic build only)
                    pt_diag = ctx->buf(8 * 8);
                    RJ_HIP(hipMemsetAsync(pt_diag->p, 0, 64, ctx->stream));
                    pp.diag = pt_diag->as<unsigned long long>();
                }
                struct DiagDump {
                    Context* ctx; BufP b; uint32_t p; uint64_t n; bool aos;
                    ~DiagDump() {
                        if (!b) return;
                        unsigned long long h[8];
                        if (hipMemcpyAsync(h, b->p, 64, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return;
                        (void)hipStreamSynchronize(ctx->stream);
                        static const char* names[] = {"clear+tail", "load issue+hash", "rank+barrier", "reserve+scan",
                                                      "stage A", "copy-out A", "stage B", "copy-out B"};
                        double tot = 0;
                        for (int i = 0; i < 8; ++i) tot += (double)h[i];
                        if (tot <= 0) return;
                        fprintf(stderr, "[rj diag] scatter pass %u over %llu tuples, phase shares (thread 0 cycles):", p, (unsigned long long)n);
                        for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.1f%%", names[i], 100.0 * h[i] / tot);
                        fprintf(stderr, "  (%.0f cycles per tile)\n", tot / ((double)n / PT_TILE));
                    }
                } diag_dump{ctx, pt_diag, p, n, false};
                if (aos_mid) {
                    const bool last = p + 1 == passes;
                    Words      o{};
                    o.w[0] = last ? waos.w[0] : MID[p % 2]->as<uint32_t>();
                    if (!last && mid_side) {
                        pp.side_out = SIDE[p % 2]->as<uint16_t>();
                        pp.next_shift = shift + pbits[p];
                        pp.next_mask = (1u << pbits[p + 1]) - 1u;
                    }
                    if (p == 0 && !ws)
                        launch_pass_scatter_src(L, src, KW, CW, pp, n_groups, o, true);
                    else if (p == 0)
                        launch_pass_scatter_dense(L, cur, P.NW, KW, pp, n_groups, o, true);
                    else
                        launch_pass_scatter_aos3(L, MID[(p - 1) % 2]->as<uint32_t>(), pp, n_groups, o.w[0]);
                    nxt = o;
                } else if (P.packed) {
                    if (packed_side && p + 1 < passes) {
                        pp.side_out = SIDE[p % 2]->as<uint16_t>();
                        pp.next_shift = shift + pbits[p];
                        pp.next_mask = (1u << pbits[p + 1]) - 1u;
                    }
                    const bool blk_out = blocked_mid && p + 1 < passes;
                    if (p == 0 && !ws)
                        launch_pass_scatter_src_packed(L, src, pp, n_groups, nxt.w[0], blk_out);
                    else if (blocked_mid)
                        launch_pass_scatter_blocked(L, cur.w[0], pp, n_groups, nxt.w[0], blk_out);
                    else
                        launch_pass_scatter_packed(L, cur.w[0], pp, n_groups, nxt.w[0]);
                } else if (p == 0 && !ws) {
                    launch_pass_scatter_src(L, src, KW, CW, pp, n_groups, nxt, P.aos3 && p + 1 == passes);
                } else {
                    launch_pass_scatter_dense(L, cur, P.NW, pair_word, pp, n_groups, nxt,
                                              P.aos3 && p + 1 == passes);
                }
                seg_off = off;
                nseg = (uint32_t)bins;
                shift += pbits[p];
                cur = nxt;
                cur_is_a = (p % 2 == 0);
                nxt = cur_is_a ? wb : wa;
                if (P.aos3 && p + 2 == passes) nxt = waos;  // the next pass is the last one
                if (aos_mid) {
                    // (MID / SIDE buffers of a pass are read by the next one only; they go back to
                    // the block cache when this function returns — all on one stream)
                }
            }
            P.w = cur;
            for (int a = 0; a < P.NW; ++a) P.wbuf[a] = cur_is_a ? A[a] : B[a];
            if (P.aos3) P.wbuf[0] = AOS;
            P.off = seg_off;
            P.NP = nseg;
            P.pbits = pbits;
            if (shape && !shape->prior_bits.empty()) P.pbits.insert(P.pbits.begin(), shape->prior_bits.begin(), shape->prior_bits.end());
            P.n_tuples = n;
            return P;
        }

        // ---------------------------------------------------------------- join
        struct Side {
            Rel*          rel = nullptr;
            uint64_t      key_col = 0;
            std::set<int> need;       // referenced columns that are not served by the key stream
            int           carry_mode = CARRY_NONE;
            int           carry_col = -1;
            int           CW = 0;
            BufP          stream;     // emitted carry stream
            int           stream_mode = ST_NONE;
            // CARRY_WIDE: the carried columns in record order (a 64-bit one first), then — if any of
            // them has NULLs — one word of validity bits (bit i = wide_cols[i] is non-NULL)
            std::vector<int> wide_cols;
            int              wide = WIDE_NONE;
            int              valid_word = -1;
            BufP             vword;  // the packed validity word per row of the child
            struct WideOut {
                BufP values, valid;
                int  mode = ST_NONE;  // dense values, or — root columns without NULLs — Page images straight away
            };
            std::map<int, WideOut> wide_out;  // per carried column: what k_split_records produced
            // A sharded join's ranks must cut their tuples alike, and whether a column needs a validity
            // word depends on the DATA of a shard: columns that hold NULLs on ANY rank are nullable on
            // every rank (bit c = column c; columns from 64 up count as nullable when `shared` is set)
            uint64_t null_mask = 0;
            bool     shared = false;
            bool     nullable(int c) const {
                if (rel->cols[c].valid != nullptr) return true;
                return shared && (c >= 64 || ((null_mask >> c) & 1u));
            }
        };
        // which of a relation's (first 64) columns hold NULLs here
        static uint64_t null_columns(const Rel& r) {
            uint64_t m = 0;
            for (size_t c = 0; c < r.cols.size() && c < 64; ++c)
                if (r.cols[c].valid != nullptr) m |= 1ull << c;
            return m;
        }

        static uint64_t pages_for(uint64_t rows, int width) {
            uint64_t rf = width == 4 ? ROWS32 : ROWS64;
            return (rows + rf - 1) / rf;
        }

        Rel empty_rel(const JoinSpec& js, Result* root_res) {
            Rel r;
            r.n = 0;
            for (size_t k = 0; k < js.out_type.size(); ++k) {
                DCol d;
                d.type = js.out_type[k];
                d.kind = COL_DENSE;
                d.width = d.type == RJ_INT32 || d.type == RJ_VARCHAR ? 4 : 8;
                r.cols.push_back(d);
            }
            if (root_res) {
                root_res->num_rows = 0;
                for (size_t k = 0; k < js.out_type.size(); ++k) {
                    ResultColumn rc;
                    rc.type = js.out_type[k];
                    root_res->cols.push_back(std::move(rc));
                }
            }
            return r;
        }

        // What a JoinNode decides before any tuple moves: key type, which columns each side
        // must deliver and how they travel (execute_hash_join, reference src/execute.cpp:266-282,
        // and the head of hash_join_omp, :43-83).
        struct JoinState {
            Side     ls, rs;
            bool     build_left = true, is_root = false, f64 = false, need_key_stream = false;
            bool     type_mismatch = false;  // probe key of another type: no row can match (:65-71)
            size_t   lw = 0, rw = 0;
            int      KW = 1;
            uint64_t cap_hint = 0;  // rows the output streams are sized for at the first attempt
            // VARCHAR join keys (hash_join_omp<std::string>, src/execute.cpp:278): each side's
            // relation is copied with one extra column — the 64-bit FNV-1a of the key strings —
            // which the join runs on; `rows` remembers where every string sits for the
            // byte-for-byte check of the joined pairs
            bool vkey = false;
            struct VKey {
                Rel            rel;
                BufP           rows, hash, valid;
                const uint8_t* pages = nullptr;
                uint32_t       n_pages = 0;
            } vk[2];  // [0] = left, [1] = right
            Side&    bs() { return build_left ? ls : rs; }
            Side&    ps() { return build_left ? rs : ls; }
        };

        // shared_nulls (sharded joins): {left, right} null_columns() OR-ed over all ranks
        void join_prepare(Rel& left, Rel& right, const JoinSpec& js, bool is_root, JoinState& st,
                          const uint64_t* shared_nulls = nullptr) {
            const size_t lw = left.cols.size(), rw = right.cols.size();
            st.lw = lw;
            st.rw = rw;
            st.is_root = is_root;
            st.build_left = js.build_left;
            if (js.left_attr >= lw || js.right_attr >= rw)
                throw_fmt(RJ_ERR_ARG, "join: key attr out of range");
            for (size_t k = 0; k < js.out_idx.size(); ++k) {
                if (js.out_idx[k] >= lw + rw) throw_fmt(RJ_ERR_ARG, "join: output attr out of range");
                const DCol& c = js.out_idx[k] < lw ? left.cols[js.out_idx[k]]
                                                   : right.cols[js.out_idx[k] - lw];
                if (c.type != js.out_type[k])
                    throw_fmt(RJ_ERR_ARG, "join: declared type differs from the child column's type");
            }
            Side &ls = st.ls, &rs = st.rs;
            ls.rel = &left;
            ls.key_col = js.left_attr;
            rs.rel = &right;
            rs.key_col = js.right_attr;
            if (shared_nulls) {
                ls.shared = rs.shared = true;
                ls.null_mask = shared_nulls[0];
                rs.null_mask = shared_nulls[1];
            }
            const DCol& bk = st.bs().rel->cols[st.bs().key_col];
            const DCol& pk = st.ps().rel->cols[st.ps().key_col];
            // KeyType = build side's key type (:271-273)
            if (bk.type < RJ_INT32 || bk.type > RJ_VARCHAR) throw_fmt(RJ_ERR_ARG, "Unsupported join type");
            // probe values of another variant alternative are never valid (:65-71)
            st.type_mismatch = pk.type != bk.type;
            st.vkey = bk.type == RJ_VARCHAR && !st.type_mismatch;
            st.KW = bk.type == RJ_INT32 ? 1 : 2;
            st.f64 = bk.type == RJ_FP64;
            // matching keys are bit-identical on both sides (FP64 included: bit-pattern equality,
            // see SrcLoader::key2), so one emitted key stream serves either side's key column.
            // Which child columns must each side deliver?
            for (size_t k = 0; k < js.out_idx.size(); ++k) {
                bool  is_left = js.out_idx[k] < lw;
                Side& s = is_left ? ls : rs;
                int   c = (int)(is_left ? js.out_idx[k] : js.out_idx[k] - lw);
                if ((uint64_t)c == s.key_col && !st.vkey)
                    st.need_key_stream = true;
                else
                    s.need.insert(c);  // (a VARCHAR key column is gathered like any other column)
            }
            for (Side* s : {&ls, &rs}) {
                if (st.vkey) {  // both row indices are needed for the string comparison of the pairs
                    s->carry_mode = CARRY_ROWIDX;
                    s->CW = 1;
                } else if (s->need.empty()) {
                    s->carry_mode = CARRY_NONE;
                    s->CW = 0;
                } else if (s->need.size() == 1 && !s->nullable(*s->need.begin())) {
                    s->carry_mode = CARRY_COLUMN;
                    s->carry_col = *s->need.begin();
                    s->CW = s->rel->cols[s->carry_col].width / 4;
                } else if (plan_wide_carry(*s, st.KW)) {
                    s->carry_mode = CARRY_WIDE;
                } else {
                    s->carry_mode = CARRY_ROWIDX;
                    s->CW = 1;
                }
            }
            st.cap_hint = std::max(left.n, right.n);
        }

        // Can the columns a side must deliver travel WITH the key (reference src/execute.cpp:236-242
        // copies any column list per output row; here up to MAX_WORDS - KW carry words do the same
        // without a row index to gather through afterwards)?  Layouts: two or three 32-bit words, or
        // a 64-bit column followed by one 32-bit word; a validity word counts as a 32-bit word.
        bool plan_wide_carry(Side& s, int KW) {
            if (!ctx->tune.wide_carry || s.need.empty()) return false;
            int  words = 0, n64 = 0;
            bool any_null = false;
            for (int c : s.need) {
                const DCol& col = s.rel->cols[c];
                if (col.width != 4 && col.width != 8) return false;
                words += col.width / 4;
                n64 += col.width == 8;
                any_null = any_null || s.nullable(c);
            }
            if (any_null) ++words;
            if (words < 2 || words > MAX_WORDS - KW || n64 > 1 || (n64 == 1 && words != 3)) return false;
            s.wide_cols.clear();
            for (int c : s.need)
                if (s.rel->cols[c].width == 8) s.wide_cols.push_back(c);
            for (int c : s.need)
                if (s.rel->cols[c].width == 4) s.wide_cols.push_back(c);
            s.wide = n64 ? WIDE_64_32 : WIDE_32S;
            s.valid_word = any_null ? words - 1 : -1;
            s.CW = words;
            return true;
        }

        // validity bits of a wide carry's columns, one word per row (k_pack_validity); call once
        // per join before the side's tuples are formed
        void prepare_wide(Side& s) {
            if (s.carry_mode != CARRY_WIDE || s.valid_word < 0 || s.rel->n == 0) return;
            const uint8_t* v[3] = {nullptr, nullptr, nullptr};
            for (size_t i = 0; i < s.wide_cols.size() && i < 3; ++i) v[i] = s.rel->cols[s.wide_cols[i]].valid;
            s.vword = ctx->buf(s.rel->n * 4);
            launch_pack_validity(L, v[0], v[1], v[2], (uint32_t)s.rel->n, s.vword->as<uint32_t>());
        }

        // radix bit plan from the build cardinality; `top_bits_taken` high hash bits are constant
        // on this rank (a sharded join's rank digit): the plan stays below them
        uint32_t join_bits(const JoinSpec& js, uint64_t build_n, uint32_t top_bits_taken = 0) {
            uint32_t bits = js.forced_bits > 0 ? (uint32_t)js.forced_bits
                                               : ceil_log2((build_n + JN_TARGET_BUILD - 1) / JN_TARGET_BUILD);
            // a third pass costs 20 B/tuple more than slightly fuller tables: stay at two passes
            // (2 * PT_MAXBITS bits) while the mean build partition still fits the LDS table with
            // a margin (rare larger partitions are joined in table-sized chunks anyway)
            if (js.forced_bits <= 0 && bits > 2 * PT_MAXBITS &&
                (build_n >> (2 * PT_MAXBITS)) <= (uint64_t)(JN_RMAX * 0.95))
                bits = 2 * PT_MAXBITS;
            // (at most 21: three passes of 7 bits — what 2^32 build rows ask for; more partitions than that
            // and the join's launch would exceed 2^32 threads.  A larger forced value is clamped.)
            bits = std::min<uint32_t>(std::max<uint32_t>(bits, 1), 21);
            if (top_bits_taken && bits > 32 - top_bits_taken) bits = 32 - top_bits_taken;
            return bits;
        }

        TupleSrc make_src(const JoinState& st, const Side& s, const JoinSpec& js) {
            TupleSrc src{};
            src.key = s.rel->cols[s.key_col].ref();
            src.n_rows = (uint32_t)s.rel->n;
            src.carry_mode = s.carry_mode;
            if (s.carry_mode == CARRY_COLUMN) {
                src.carry = s.rel->cols[s.carry_col].ref();
                // a base table's row-id column (VARCHAR stand-in) IS the row index
                if (src.carry.kind == COL_IOTA) src.carry_mode = CARRY_ROWIDX;
            }
            if (s.carry_mode == CARRY_WIDE) {
                ColRef refs[3] = {};
                size_t k = 0;
                for (int c : s.wide_cols) refs[k++] = s.rel->cols[c].ref();
                if (s.valid_word >= 0)
                    refs[k++] = ColRef{s.vword ? s.vword->as<uint8_t>() : nullptr, nullptr, COL_DENSE, 4};
                src.wide = s.wide;
                src.carry = refs[0];
                src.carry2 = refs[1];
                src.carry3 = refs[2];
            }
            src.key_f64 = st.f64 ? 1 : 0;
            src.prehashed = js.prehashed ? 1 : 0;
            return src;
        }

        // execute_hash_join + hash_join_omp (reference src/execute.cpp:43-282) on one device
        Rel join_core(Rel& left, Rel& right, const JoinSpec& js, Result* root_res) {
            // with an empty child the reference returns {} before looking at anything (:50)
            if (left.n == 0 || right.n == 0) return empty_rel(js, root_res);
            JoinState st;
            join_prepare(left, right, js, root_res != nullptr, st);
            if (st.type_mismatch) return empty_rel(js, root_res);
            if (st.vkey) hash_varchar_keys(st);
            prepare_wide(st.ls);
            prepare_wide(st.rs);
            Side&          bs = st.bs();
            Side&          ps = st.ps();
            const uint32_t bits = join_bits(js, bs.rel->n, js.top_bits_taken);
            if (ctx->tune.diag >= 2)
                fprintf(stderr, "[rj diag] join build=%llu probe=%llu bits=%u cw=%d/%d\n",
                        (unsigned long long)bs.rel->n, (unsigned long long)ps.rel->n, bits, bs.CW, ps.CW);
            // A build side that fits one LDS table is not partitioned at all: every workgroup builds
            // the same table and streams a slice of the probe child past it (k_join_bcast)
            const bool bcast = bs.rel->n <= (uint64_t)JN_RMAX && js.forced_bits <= 0 && !js.prehashed &&
                               ctx->tune.bcast != 0;
            Parted PB, PP;
            if (!bcast) {
                TupleSrc sb = make_src(st, bs, js), sp = make_src(st, ps, js);
                PB = partition(&sb, nullptr, st.KW, bs.CW, bits);
                PP = partition(&sp, nullptr, st.KW, ps.CW, bits);
            }
            return join_finish(st, js, bcast ? nullptr : &PB, bcast ? nullptr : &PP, bits, root_res);
        }

        // Build + probe over co-partitioned tuples (PB / PP; nullptr = broadcast join straight
        // from the children's columns), output streams, late materialisation, result pages.
        Rel join_finish(JoinState& st, const JoinSpec& js, const Parted* PBp, const Parted* PPp,
                        uint32_t bits, Result* root_res) {
            Side &         ls = st.ls, &rs = st.rs, &bs = st.bs(), &ps = st.ps();
            const size_t   lw = st.lw;
            const bool     is_root = st.is_root, bcast = PBp == nullptr;
            const int      KW = st.KW;
            const bool     need_key_stream = st.need_key_stream;
            const uint64_t probe_n = bcast ? ps.rel->n : (uint64_t)0;
            (void)probe_n;
            uint32_t max_tasks = 0;
            BufP     tasks;
            BufP     counters = ctx->buf(16);  // [0..7] out cursor (u64), [8..11] n_heavy
            if (!bcast) {
                const Parted &PB = *PBp, &PP = *PPp;
                // heavy probe partitions -> task list
                max_tasks = (uint32_t)(2 * (PP.n_tuples / JN_HEAVY) + 2);
                tasks = ctx->buf((uint64_t)max_tasks * 12);
                launch_heavy_tasks_zeroed(PB, PP, tasks, counters, max_tasks);
            }

            // stream destinations
            auto stream_mode = [&](int width, bool direct_output) -> int {
                if (is_root && direct_output) return width == 4 ? ST_PAGED32 : ST_PAGED64;
                return width == 4 ? ST_DENSE32 : ST_DENSE64;
            };
            auto stream_bytes = [&](int mode, uint64_t rows) -> uint64_t {
                switch (mode) {
                case ST_DENSE32: return rows * 4;
                case ST_DENSE64: return rows * 8;
                case ST_PAGED32: return pages_for(rows, 4) * PAGE_BYTES;
                case ST_PAGED64: return pages_for(rows, 8) * PAGE_BYTES;
                case ST_DENSE96: return rows * 12;
                default: return 0;
                }
            };
            int key_mode = need_key_stream ? stream_mode(KW * 4, true) : ST_NONE;
            for (Side* s : {&ls, &rs}) {
                if (s->carry_mode == CARRY_NONE)
                    s->stream_mode = ST_NONE;
                else if (s->carry_mode == CARRY_ROWIDX)
                    s->stream_mode = ST_DENSE32;
                else if (s->carry_mode == CARRY_WIDE)
                    s->stream_mode = s->CW == 2 ? ST_DENSE64 : ST_DENSE96;  // records, split below
                else {
                    const DCol& c = s->rel->cols[s->carry_col];
                    s->stream_mode = stream_mode(c.width, c.type != RJ_VARCHAR);
                }
            }

            uint64_t cap = st.cap_hint;
            cap = std::min<uint64_t>(cap + 1024, 0xfffffff0ull);
            BufP            key_stream;
            uint64_t        nrows = 0;
            std::set<void*> finished;  // paged buffers that already got their headers
            for (int attempt = 0; attempt < 2; ++attempt) {
                key_stream = key_mode != ST_NONE ? ctx->buf(stream_bytes(key_mode, cap)) : BufP();
                for (Side* s : {&ls, &rs})
                    s->stream = s->stream_mode != ST_NONE ? ctx->buf(stream_bytes(s->stream_mode, cap))
                                                          : BufP();
                RJ_HIP(hipMemsetAsync(counters->p, 0, 8, ctx->stream));
                if (bcast) {
                    BcastParams bp{};
                    bp.R = make_src(st, bs, js);
                    bp.S = make_src(st, ps, js);
                    bp.key = OutStream{key_stream ? key_stream->as<uint8_t>() : nullptr, key_mode, 0};
                    bp.bc = OutStream{bs.stream ? bs.stream->as<uint8_t>() : nullptr, bs.stream_mode, 0};
                    bp.pc = OutStream{ps.stream ? ps.stream->as<uint8_t>() : nullptr, ps.stream_mode, 0};
                    bp.out_cursor = counters->as<unsigned long long>();
                    bp.out_cap = cap;
                    const uint64_t chunks = (ps.rel->n + JN_SUB - 1) / JN_SUB;
                    launch_join_bcast(L, KW, bs.CW, ps.CW, bp,
                                      (uint32_t)std::min<uint64_t>(chunks, (uint64_t)ctx->compute_units() * 8));
                } else {
                const Parted &PB = *PBp, &PP = *PPp;
                JoinParams jp{};
                jp.R = PB.w;
                jp.S = PP.w;
                jp.packR = PB.packed ? 1 : 0;
                jp.packS = PP.packed ? 1 : 0;
                jp.aosR = PB.aos3 ? 1 : 0;
                jp.aosS = PP.aos3 ? 1 : 0;
                jp.offR = PB.off->as<uint32_t>();
                jp.offS = PP.off->as<uint32_t>();
                jp.NP = PB.NP;
                jp.radix_bits = bits;
                jp.n_pass = (uint32_t)PB.pbits.size();
                for (size_t i = 0; i < PB.pbits.size() && i < 4; ++i) jp.pass_bits[i] = PB.pbits[i];
                jp.key = OutStream{key_stream ? key_stream->as<uint8_t>() : nullptr, key_mode, 0};
                jp.bc = OutStream{bs.stream ? bs.stream->as<uint8_t>() : nullptr, bs.stream_mode, 0};
                jp.pc = OutStream{ps.stream ? ps.stream->as<uint8_t>() : nullptr, ps.stream_mode, 0};
                jp.out_cursor = counters->as<unsigned long long>();
                jp.out_cap = cap;
                jp.heavy_tasks = tasks->as<uint32_t>();
                jp.n_heavy = counters->as<uint32_t>() + 2;
                BufP diag;
                if (ctx->tune.diag) {
                    diag = ctx->buf(16 * 8);
                    RJ_HIP(hipMemsetAsync(diag->p, 0, 16 * 8, ctx->stream));
                    jp.diag = diag->as<unsigned long long>();
                }
                // one launch: heavy-task workgroups first, then one workgroup per partition
                jp.heavy_grid = max_tasks;
                const uint32_t ppw = join_partitions_per_workgroup(KW, bs.CW, jp);
                launch_join(L, KW, bs.CW, ps.CW, jp, max_tasks + (PB.NP + ppw - 1) / ppw);
                if (diag) {
                    unsigned long long hd[16];
                    RJ_HIP(hipMemcpyAsync(hd, diag->p, sizeof hd, hipMemcpyDeviceToHost, ctx->stream));
                    ctx->sync();
                    static const char* names[] = {"loads issued", "table clear", "build", "count/probe",
                                                  "prefix+barrier", "reserve", "emit"};
                    double tot = 0;
                    for (int i = 0; i < 7; ++i) tot += (double)hd[i];
                    fprintf(stderr, "[rj diag] join phases (cycles of thread 0, summed over %u workgroups):\n", PB.NP);
                    for (int i = 0; i < 7; ++i)
                        fprintf(stderr, "[rj diag]   %-16s %6.1f %%  %8.0f cyc/wg\n", names[i],
                                100.0 * hd[i] / tot, (double)hd[i] / PB.NP);
                    jp.diag = nullptr;
                }
            }
            // Page headers of the streams the probe wrote straight into Page images: done on
            // the device from the device-side row count, so nothing waits for the read-back
            finished.clear();
            {
                uint8_t* fp[3];
                int      fw[3];
                uint32_t nf = 0;
                auto add = [&](const BufP& b, int mode) {
                    if (!b || (mode != ST_PAGED32 && mode != ST_PAGED64) || finished.count(b->p)) return;
                    fp[nf] = b->as<uint8_t>();
                    fw[nf] = mode == ST_PAGED32 ? 4 : 8;
                    ++nf;
                    finished.insert(b->p);
                };
                add(key_stream, key_mode);
                add(ls.stream, ls.stream_mode);
                add(rs.stream, rs.stream_mode);
                launch_finish_streams(L, fp, fw, nf, counters->as<unsigned long long>(), cap);
            }
            unsigned long long h = 0;
            RJ_HIP(hipMemcpyAsync(&h, counters->p, 8, hipMemcpyDeviceToHost, ctx->stream));
            ctx->sync();
            nrows = h;
            if (nrows <= cap) break;
            if (nrows > 0xfffffff0ull)
                throw_fmt(RJ_ERR_UNSUPPORTED, "join result exceeds 2^32 rows (%llu)", h);
            if (attempt == 1) throw_fmt(RJ_ERR_DEVICE, "join output overflowed twice");
            cap = nrows;  // exact size, run the probe again
        }

        if (st.vkey && nrows) nrows = verify_varchar_pairs(st, nrows);

        // wide carries: the emitted records -> one dense array (+ validity bytes) per column
        for (Side* s : {&ls, &rs}) {
            if (s->carry_mode != CARRY_WIDE) continue;
            SplitParams sp{};
            sp.rec = s->stream->as<uint32_t>();
            sp.cw = (uint32_t)s->CW;
            sp.valid_word = s->valid_word;
            int word = 0;
            for (size_t i = 0; i < s->wide_cols.size(); ++i) {
                const DCol&   col = s->rel->cols[s->wide_cols[i]];
                Side::WideOut wo;
                // a root column without NULLs goes into its Page images here (the header words are
                // filled in below), anything else into a dense array
                const bool nullable = s->nullable(s->wide_cols[i]);  // (here, or on another rank of a sharded join)
                const bool to_pages = is_root && !nullable && col.type != RJ_VARCHAR;  // (a VARCHAR column travels as row ids)
                wo.mode = to_pages ? (col.width == 4 ? ST_PAGED32 : ST_PAGED64) : (col.width == 4 ? ST_DENSE32 : ST_DENSE64);
                wo.values = ctx->buf(stream_bytes(wo.mode, std::max<uint64_t>(nrows, 1)));
                if (nullable) wo.valid = ctx->buf(std::max<uint64_t>(nrows, 1));
                sp.col[i].out = wo.values->as<uint8_t>();
                sp.col[i].valid = wo.valid ? wo.valid->as<uint8_t>() : nullptr;
                sp.col[i].word = word;
                sp.col[i].width = col.width;
                sp.col[i].valid_bit = (int32_t)i;
                sp.col[i].paged = to_pages ? 1 : 0;
                word += col.width / 4;
                s->wide_out[s->wide_cols[i]] = wo;
            }
            sp.n_cols = (int32_t)s->wide_cols.size();
            launch_split_records(L, sp, nrows);
        }

        // ------------------------------------------------ assemble the outputs
        Rel out;
        out.n = nrows;
        if (is_root) root_res->num_rows = nrows;
        for (size_t k = 0; k < js.out_idx.size(); ++k) {
            bool        is_left = js.out_idx[k] < lw;
            Side&       s = is_left ? ls : rs;
            int         c = (int)(is_left ? js.out_idx[k] : js.out_idx[k] - lw);
            const DCol& src = s.rel->cols[c];
            DCol        d;
            d.type = src.type;
            d.width = src.width;
            d.kind = COL_DENSE;
            d.vc_table = src.vc_table;
            d.vc_col = src.vc_col;
            BufP buf;           // where this column's values/pages live
            int  buf_mode = ST_NONE;
            BufP valid;
            if ((uint64_t)c == s.key_col) {
                buf = key_stream;
                buf_mode = key_mode;
            } else if (s.carry_mode == CARRY_COLUMN) {
                buf = s.stream;
                buf_mode = s.stream_mode;
            } else if (s.carry_mode == CARRY_WIDE) {
                const Side::WideOut& wo = s.wide_out.at(c);
                buf = wo.values;
                valid = wo.valid;
                buf_mode = wo.mode;
            } else {
                // generic path: gather the child column through the row-index stream
                const uint32_t* idx = s.stream->as<uint32_t>();
                if (src.kind == COL_IOTA) {
                    buf = s.stream;  // row ids of the base table ARE the stream
                    buf_mode = ST_DENSE32;
                } else {
                    bool to_pages = is_root && src.type != RJ_VARCHAR && src.valid == nullptr;
                    buf_mode = to_pages ? (src.width == 4 ? ST_PAGED32 : ST_PAGED64)
                                        : (src.width == 4 ? ST_DENSE32 : ST_DENSE64);
                    buf = ctx->buf(stream_bytes(buf_mode, std::max<uint64_t>(nrows, 1)));
                    if (src.valid) valid = ctx->buf(std::max<uint64_t>(nrows, 1));
                    launch_gather(L, src.ref(), idx, nrows,
                                  OutStream{buf->as<uint8_t>(), buf_mode, 0},
                                  valid ? valid->as<uint8_t>() : nullptr);
                }
            }
            if (!is_root) {
                d.hold = buf;
                d.ptr = buf ? buf->as<uint8_t>() : nullptr;
                d.hold_valid = valid;
                d.valid = valid ? valid->as<uint8_t>() : nullptr;
                out.cols.push_back(d);
                continue;
            }
            // ---- root: Page images (replaces Table::to_columnar, build_table.cpp:456-681)
            ResultColumn rc;
            rc.type = src.type;
            if (nrows == 0) {
                // empty result: typed columns with zero pages (reference tests/unit_tests.cpp:24-27)
            } else if (src.type == RJ_VARCHAR) {
                varchar_root(buf->as<uint32_t>(), nrows, src, rc);
            } else if (buf_mode == ST_PAGED32 || buf_mode == ST_PAGED64) {
                if (!finished.count(buf->p)) {
                    launch_finish_pages(L, buf->as<uint8_t>(), nrows, src.width);
                    finished.insert(buf->p);
                }
                rc.dev_pages = buf;
                rc.n_pages = pages_for(nrows, src.width);
            } else {
                // dense values (+ validity): encode pages on the device
                rc.n_pages = pages_for(nrows, src.width);
                rc.dev_pages = ctx->buf(rc.n_pages * PAGE_BYTES);
                if (valid) {
                    launch_encode_nullable(L, buf->as<uint8_t>(), valid->as<uint8_t>(), nrows,
                                           src.width, rc.dev_pages->as<uint8_t>());
                } else {
                    DCol dense;
                    dense.kind = COL_DENSE;
                    dense.width = src.width;
                    dense.ptr = buf->as<uint8_t>();
                    launch_gather(L, dense.ref(), nullptr, nrows,
                                  OutStream{rc.dev_pages->as<uint8_t>(),
                                            src.width == 4 ? ST_PAGED32 : ST_PAGED64, 0},
                                  nullptr);
                    launch_finish_pages(L, rc.dev_pages->as<uint8_t>(), nrows, src.width);
                }
            }
            root_res->cols.push_back(std::move(rc));
        }
        return out;
    }

    // The VARCHAR pages of a base column + their row directory in HBM (uploaded once per table).
    struct VcDev {
        const uint8_t*  pages;
        uint32_t        n_pages;
        const uint32_t* dir;
    };
    VcDev ensure_vc_dev(int vc_table, int vc_col) {
        if (vc_table < 0 || (uint64_t)vc_table >= n_tables) throw_fmt(RJ_ERR_ARG, "VARCHAR column without provenance");
        const Table*       t = table_by_id((uint64_t)vc_table);
        const TableColumn& tc = t->cols[vc_col];
        if (tc.vc_pages.size() > 0xfffffff0ull || t->num_rows > 0xfffffff0ull)
            throw_fmt(RJ_ERR_UNSUPPORTED, "VARCHAR column too large for the device path");
        auto key = std::make_pair(vc_table, vc_col);
        auto it = vc_dir_.find(key);
        if (it == vc_dir_.end()) {
            it = vc_dir_.emplace(key, std::vector<uint64_t>()).first;
            varchar_dir_build(tc.vc_pages.data(), tc.vc_pages.size(), t->num_rows, it->second);
        }
        const uint32_t npg = (uint32_t)tc.vc_pages.size();
        if (!tc.vc_dev) {  // (a table ingested on the device brings its pages along: rj_ingest.hip)
            tc.vc_dev = ctx->buf(std::max<uint64_t>(npg, 1) * PAGE_BYTES);
            upload_host_pages(ctx, tc.vc_pages.data(), npg, tc.vc_dev->as<uint8_t>());
        }
        if (!tc.vc_dev_dir) {
            std::vector<uint32_t> dir32(it->second.begin(), it->second.end());
            tc.vc_dev_dir = ctx->buf(dir32.size() * 4);
            RJ_HIP(hipMemcpyAsync(tc.vc_dev_dir->p, dir32.data(), dir32.size() * 4, hipMemcpyHostToDevice,
                                  ctx->stream));
            ctx->sync();  // dir32 is a local
        }
        return VcDev{tc.vc_dev->as<uint8_t>(), npg, tc.vc_dev_dir->as<uint32_t>()};
    }

    // VARCHAR join keys: hash both key columns, join on the hashes (see JoinState::VKey)
    void hash_varchar_keys(JoinState& st) {
        int k = 0;
        for (Side* s : {&st.ls, &st.rs}) {
            JoinState::VKey& v = st.vk[k++];
            const DCol       kc = s->rel->cols[s->key_col];  // a row-id column of its base table
            if (kc.kind != COL_IOTA && kc.kind != COL_DENSE) throw_fmt(RJ_ERR_ARG, "VARCHAR key column of unknown shape");
            const VcDev d = ensure_vc_dev(kc.vc_table, kc.vc_col);
            const uint32_t n = (uint32_t)s->rel->n;
            v.pages = d.pages;
            v.n_pages = d.n_pages;
            v.rows = ctx->buf(std::max<uint64_t>(n, 1) * sizeof(VcRow));
            v.hash = ctx->buf(std::max<uint64_t>(n, 1) * 8);
            v.valid = ctx->buf(std::max<uint64_t>(n, 1));
            launch_vc_hash(L, d.pages, d.n_pages, d.dir,
                           kc.kind == COL_DENSE ? reinterpret_cast<const uint32_t*>(kc.ptr) : nullptr, n,
                           v.rows->as<VcRow>(), v.hash->as<uint64_t>(), v.valid->as<uint8_t>(),
                           ctx->tune.vkey_hash_bits > 0 && ctx->tune.vkey_hash_bits < 64
                               ? ((1ull << ctx->tune.vkey_hash_bits) - 1ull)
                               : ~0ull);
            v.rel = *s->rel;
            DCol h;
            h.type = RJ_INT64;
            h.kind = COL_DENSE;
            h.width = 8;
            h.ptr = v.hash->as<uint8_t>();
            h.valid = v.valid->as<uint8_t>();
            v.rel.cols.push_back(h);
            s->rel = &v.rel;
            s->key_col = v.rel.cols.size() - 1;
        }
    }

    // Every joined pair is compared byte for byte; the (build row, probe row) streams are
    // compacted if — and only if — a hash collision let a pair of different strings through.
    uint64_t verify_varchar_pairs(JoinState& st, uint64_t nrows) {
        Side &                 bs = st.bs(), &ps = st.ps();
        const JoinState::VKey& vb = st.vk[st.build_left ? 0 : 1];
        const JoinState::VKey& vp = st.vk[st.build_left ? 1 : 0];
        const uint32_t         n = (uint32_t)nrows;
        BufP                   keep = ctx->buf(n), bad = ctx->buf(16);
        RJ_HIP(hipMemsetAsync(bad->p, 0, 16, ctx->stream));
        launch_vc_verify(L, vb.pages, vb.n_pages, vb.rows->as<VcRow>(), vp.pages, vp.n_pages, vp.rows->as<VcRow>(),
                         bs.stream->as<uint32_t>(), ps.stream->as<uint32_t>(), n, keep->as<uint8_t>(),
                         bad->as<unsigned long long>());
        unsigned long long n_bad = 0;
        RJ_HIP(hipMemcpyAsync(&n_bad, bad->p, 8, hipMemcpyDeviceToHost, ctx->stream));
        ctx->sync();
        if (n_bad == 0) return nrows;
        BufP nb = ctx->buf(std::max<uint64_t>(nrows - n_bad, 1) * 4), np = ctx->buf(std::max<uint64_t>(nrows - n_bad, 1) * 4);
        launch_vc_compact(L, keep->as<uint8_t>(), bs.stream->as<uint32_t>(), ps.stream->as<uint32_t>(), n,
                          nb->as<uint32_t>(), np->as<uint32_t>(), bad->as<unsigned long long>() + 1);
        bs.stream = nb;
        ps.stream = np;
        return nrows - n_bad;
    }

    void launch_heavy_tasks_zeroed(const Parted& PB, const Parted& PP, const BufP& tasks,
                                   const BufP& counters, uint32_t max_tasks) {
        RJ_HIP(hipMemsetAsync(counters->p, 0, 16, ctx->stream));
        launch_heavy_tasks(L, PB.off->as<uint32_t>(), PP.off->as<uint32_t>(), PB.NP,
                           tasks->as<uint32_t>(), counters->as<uint32_t>() + 2, max_tasks);
    }

    // VARCHAR at the root: row ids -> host, strings gathered from the base table's
    // pages and encoded with the reference's fill rule (build_table.cpp:595-677).
    void varchar_root(const uint32_t* dev_rowids, uint64_t n, const DCol& src, ResultColumn& rc) {
        if (src.vc_table < 0 || (uint64_t)src.vc_table >= n_tables)
            throw_fmt(RJ_ERR_ARG, "VARCHAR column without provenance");
        const Table*       t = table_by_id((uint64_t)src.vc_table);
        const TableColumn& tc = t->cols[src.vc_col];
        const bool diag = ctx->tune.diag >= 2;
        auto       tv0 = std::chrono::steady_clock::now();
        auto       key = std::make_pair(src.vc_table, src.vc_col);
        if (ctx->tune.varchar_dev_rows > 0 && n >= (uint64_t)ctx->tune.varchar_dev_rows &&
            tc.vc_pages.size() <= 0xfffffff0ull && t->num_rows <= 0xfffffff0ull) {
            // ---- large result: gather + encode on the device (rj_varchar_dev.hip)
            const VcDev    vd = ensure_vc_dev(src.vc_table, src.vc_col);
            const uint32_t npg = vd.n_pages;
            auto           tv1 = std::chrono::steady_clock::now();
            const uint32_t nr = (uint32_t)n, chunks = (nr + VC_CHUNK - 1) / VC_CHUNK;
            BufP           vrows = ctx->buf((uint64_t)nr * sizeof(VcRow));
            BufP           pcnt = ctx->buf((uint64_t)chunks * 4), pbase = ctx->buf(((uint64_t)chunks + 1) * 4);
            launch_vc_resolve(L, vd.pages, npg, vd.dir, dev_rowids, nr, vrows->as<VcRow>());
            launch_vc_walk(L, vrows->as<VcRow>(), nr, pcnt->as<uint32_t>(), nullptr, nullptr);
            launch_scan_bins(L, pcnt->as<uint32_t>(), chunks, pbase->as<uint32_t>(), nullptr);
            uint32_t n_out = 0;
            RJ_HIP(hipMemcpyAsync(&n_out, pbase->as<uint32_t>() + chunks, 4, hipMemcpyDeviceToHost, ctx->stream));
            ctx->sync();
            BufP plist = ctx->buf(std::max<uint64_t>(n_out, 1) * sizeof(VcPage));
            rc.dev_pages = ctx->buf(std::max<uint64_t>(n_out, 1) * PAGE_BYTES);
            launch_vc_walk(L, vrows->as<VcRow>(), nr, pcnt->as<uint32_t>(), pbase->as<uint32_t>(),
                           plist->as<VcPage>());
            launch_vc_encode(L, vd.pages, npg, vrows->as<VcRow>(), plist->as<VcPage>(), n_out,
                             rc.dev_pages->as<uint8_t>());
            rc.n_pages = n_out;
            if (diag) {
                ctx->sync();
                auto tv2 = std::chrono::steady_clock::now();
                auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
                fprintf(stderr, "[rj host]   varchar col on the device: %llu rows -> %u pages, source pages in HBM %.2f ms, resolve+walk+encode %.2f ms\n",
                        (unsigned long long)n, n_out, ms(tv0, tv1), ms(tv1, tv2));
            }
            return;
        }
        std::vector<uint32_t> ids(n);
        RJ_HIP(hipMemcpyAsync(ids.data(), dev_rowids, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        ctx->sync();
        auto tv1 = std::chrono::steady_clock::now();
        auto it = vc_dir_.find(key);
        if (it == vc_dir_.end()) {
            it = vc_dir_.emplace(key, std::vector<uint64_t>()).first;
            varchar_dir_build(tc.vc_pages.data(), tc.vc_pages.size(), t->num_rows, it->second);
        }
        auto tv2 = std::chrono::steady_clock::now();
        varchar_gather_encode(tc.vc_pages.data(), tc.vc_pages.size(), it->second, ids.data(), n,
                              rc.host_pages, rc.n_pages);
        if (diag) {
            auto tv3 = std::chrono::steady_clock::now();
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            fprintf(stderr, "[rj host]   varchar col: %llu rows, wait+D2H %.2f ms, page directory %.2f ms, gather+encode %.2f ms\n",
                    (unsigned long long)n, ms(tv0, tv1), ms(tv1, tv2), ms(tv2, tv3));
        }
    }
};


// ================================================================ sharded executor
// One JoinNode across the ranks of a job: every rank
//   stage A   forms (hashed key, carry) tuples of its shard of both children and partitions them
//             by (OWNER rank = top log2(world) hash bits, first local digit = low bits) in ONE radix
//             pass (fan-out = world x 2^s <= 512; RJ_TUNE_FOLD_OWNER=0: by owner only);
//   exchange  ONE variable-size all-to-all per relation (rj_comm: peer copies or RCCL), on its own
//             stream — the build side's exchange overlaps the probe side's stage A, the probe side's
//             the build side's stage B; its layout is host arithmetic over the all-gathered counts
//             (rj_xplan.cpp);
//   stage B   runs the remaining radix passes + build/probe on what arrived (the runs of a digit as
//             input segments of its first pass), i.e. the single-device join on an N-th of the data.
// Results stay on the owning rank (no reduction).  One host thread drives all local ranks phase by
// phase; everything between the count read-backs is asynchronous.  What a rank decides from its own
// DATA and that shapes what is exchanged (which columns need a validity word) is agreed on first;
// local failures travel as status words with the counts, and every wait on a peer is bounded.
class ShardedExec {
   public:
    ShardedExec(Context* group, const rj_plan* plan, Table* const* tables, uint64_t n_inputs, int flags)
        : g_(group), plan_(plan), nl_(group->n_lanes()) {
        if (!plan || plan->root >= plan->n_nodes) throw_fmt(RJ_ERR_ARG, "bad plan root");
        comm_ = group->comm.get();
        world_ = comm_ ? comm_->world() : 1;
        rank_base_ = comm_ ? comm_->rank_base() : 0;
        rb_ = ceil_log2((uint64_t)world_);
        for (int l = 0; l < nl_; ++l) {
            Context* c = group->lane(l);
            for (uint64_t i = 0; i < n_inputs; ++i) {
                Table* t = tables[(size_t)l * n_inputs + i];
                if (!t) throw_fmt(RJ_ERR_ARG, "null table");
                if (t->ctx != c) throw_fmt(RJ_ERR_ARG, "table %llu of device %d lives on another device's context",
                                           (unsigned long long)i, l);
            }
            ex_.emplace_back(new Exec(c, plan, tables + (size_t)l * n_inputs, n_inputs, flags));
        }
    }

    void run(Result** out) {
        std::vector<std::unique_ptr<Result>> res;
        for (int l = 0; l < nl_; ++l) {
            res.emplace_back(new rj_result());
            res.back()->ctx = g_->lane(l);
        }
        std::vector<Result*> rp;
        for (auto& r : res) rp.push_back(r.get());
        const rj_node& root = plan_->nodes[plan_->root];
        try {
            if (root.kind == RJ_NODE_SCAN) {
                for (int l = 0; l < nl_; ++l) {
                    use(l);
                    ex_[l]->root_scan(root, *res[l]);
                }
            } else {
                (void)node(plan_->root, &rp, 0);
            }
            for (const LocalErr& e : pending_)  // (a failure of the last join's local part)
                if (e.code) throw rj::Error(e.code, e.msg);
            sync_all();
        } catch (...) {
            // kernels and copies still queued may reference buffers about to be released
            // (compute streams never wait on an unfinished exchange — see stage B — so they drain;
            // the exchange streams are waited for with the transport's bound, unless it gave up)
            for (int l = 0; l < nl_; ++l) {
                (void)hipSetDevice(g_->lane(l)->device);
                (void)hipStreamSynchronize(g_->lane(l)->stream);
                if (comm_ && !comm_->failed()) {
                    try {
                        comm_->wait_stream(l, "the exchange streams (after an error)");
                    } catch (...) {
                    }
                }
            }
            (void)hipSetDevice(g_->device);
            throw;
        }
        use(0);
        for (int l = 0; l < nl_; ++l) out[l] = res[l].release();
    }

   private:
    Context*                           g_;
    const rj_plan*                     plan_;
    int                                nl_, world_ = 1, rank_base_ = 0;
    uint32_t                           rb_ = 0;
    Comm*                              comm_ = nullptr;
    std::vector<std::unique_ptr<Exec>> ex_;

    void use(int l) { RJ_HIP(hipSetDevice(g_->lane(l)->device)); }
    void sync_all() {
        for (int l = 0; l < nl_; ++l) {
            use(l);
            g_->lane(l)->sync();
        }
    }

    struct Ev {  // RAII event on a lane's device
        hipEvent_t e = nullptr;
        Ev() = default;
        Ev(const Ev&) = delete;
        Ev& operator=(const Ev&) = delete;
        void make() { RJ_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
        ~Ev() {
            if (e) (void)hipEventDestroy(e);
        }
    };

    std::vector<Rel> node(uint64_t idx, std::vector<Result*>* root_res, int depth) {
        if (idx >= plan_->n_nodes) throw_fmt(RJ_ERR_ARG, "bad node index");
        if (depth > 4096) throw_fmt(RJ_ERR_ARG, "plan too deep (cycle?)");
        const rj_node& n = plan_->nodes[idx];
        if (n.kind == RJ_NODE_SCAN) {
            // a scan that fails on one rank (its shard's pages are malformed: "row_idx") must not
            // leave the other ranks waiting in the next join's collectives: the error is kept and
            // travels as that join's first status word
            std::vector<Rel> r((size_t)nl_);
            if (pending_.empty()) pending_.resize((size_t)nl_);
            for (int l = 0; l < nl_; ++l)
                guarded(pending_[l], [&] {
                    use(l);
                    inject_failure(5, l);
                    r[l] = ex_[l]->scan(n);
                });
            return r;
        }
        if (n.kind != RJ_NODE_JOIN) throw_fmt(RJ_ERR_ARG, "bad node kind");
        std::vector<Rel> L = node(n.left, nullptr, depth + 1);
        std::vector<Rel> R = node(n.right, nullptr, depth + 1);
        JoinSpec         js;
        js.build_left = n.build_left != 0;
        js.left_attr = n.left_attr;
        js.right_attr = n.right_attr;
        js.out_idx.assign(n.out_idx, n.out_idx + n.n_out);
        js.out_type.assign(n.out_type, n.out_type + n.n_out);
        js.forced_bits = g_->radix_bits_override;
        return join(L, R, js, root_res);
    }

    // bytes per tuple of array `a` in the partition layout, 0 = no such array
    static uint32_t array_width(const Parted& P, int KW, int CW, int a) {
        if (P.packed) return a == 0 ? 8u : 0u;
        if (a >= P.NW) return 0;
        const int pw = CW >= 2 ? KW + CW - 2 : -1;  // the last two carry words are one pair array
        if (pw >= 0 && a == pw + 1) return 0;
        return a == pw ? 8u : 4u;
    }

    // A local failure on one rank must not leave its peers blocked in a collective: every rank
    // reports a status word with the counts it all-gathers anyway, and ALL ranks give up together,
    // before any data moves.  The failing rank rethrows its own error, the others name it.
    struct LocalErr {
        int         code = 0;
        std::string msg;
    };
    std::vector<LocalErr> pending_;  // per local rank: a scan's failure, waiting for the next join's status word
    template <class F>
    static void guarded(LocalErr& e, F&& f) {
        if (e.code) return;  // this rank failed earlier: it only keeps the collectives company
        try {
            f();
        } catch (const rj::Error& x) {
            e.code = x.code ? x.code : RJ_ERR_DEVICE;
            e.msg = x.what();
        } catch (const std::bad_alloc&) {
            e.code = RJ_ERR_NOMEM;
            e.msg = "host allocation failed";
        } catch (const std::exception& x) {
            e.code = RJ_ERR_DEVICE;
            e.msg = x.what();
        }
    }
    void agree(const std::vector<std::vector<uint64_t>>& all, size_t status_at, const std::vector<LocalErr>& lerr,
               const char* doing) {
        int bad = -1;
        for (int r = 0; r < world_ && bad < 0; ++r)
            if (all[r][status_at] != 0) bad = r;
        if (bad < 0) return;
        for (int l = 0; l < nl_; ++l)
            if (lerr[l].code) throw rj::Error(lerr[l].code, lerr[l].msg);
        throw_fmt(RJ_ERR_DEVICE, "sharded join: rank %d failed (status %llu) while %s; every rank gives up before the exchange",
                  bad, (unsigned long long)all[bad][status_at], doing);
    }
    void inject_failure(int at, int lane) {
        if (g_->tune.debug_shard_fail == at && g_->tune.debug_shard_fail_rank == rank_base_ + lane)
            throw_fmt(RJ_ERR_NOMEM, "injected failure %d on rank %d (RJ_DEBUG_SHARD_FAIL)", at, rank_base_ + lane);
    }
    void wait_xfer(int l, const char* what) {
        if (comm_) comm_->wait_stream(l, what);
    }

    std::vector<Rel> join(std::vector<Rel>& left, std::vector<Rel>& right, const JoinSpec& js,
                          std::vector<Result*>* root_res) {
        std::vector<Exec::JoinState> st((size_t)nl_);
        const bool diag = g_->tune.diag >= 2;
        auto       t_start = std::chrono::steady_clock::now();
        auto       lap = [&](const char* what) {
            if (!diag) return;
            sync_all();
            for (int l = 0; l < nl_; ++l) wait_xfer(l, what);
            auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "[rj sharded] rank %d: %-28s %8.2f ms\n", rank_base_, what,
                    std::chrono::duration<double, std::milli>(now - t_start).count());
            t_start = now;
        };
        // ---- what every rank decides locally, then agrees on globally
        std::vector<LocalErr>              lerr((size_t)nl_);
        if (!pending_.empty()) lerr = pending_;  // (a scan below this join failed on a local rank)
        std::vector<std::vector<uint64_t>> mine((size_t)nl_), all;
        // which columns hold NULLs on ANY rank: the carry layout (a validity word or not, one column
        // as it is or value + validity) follows the data, and every rank must pick the same one
        uint64_t shared_nulls[2] = {0, 0};
        for (int l = 0; l < nl_; ++l) mine[l] = {Exec::null_columns(left[l]), Exec::null_columns(right[l])};
        gather(mine, 2, all);
        for (int r = 0; r < world_; ++r) {
            shared_nulls[0] |= all[r][0];
            shared_nulls[1] |= all[r][1];
        }
        for (int l = 0; l < nl_; ++l) {
            bool ok = true;
            guarded(lerr[l], [&] {
                use(l);
                inject_failure(1, l);
                ex_[l]->join_prepare(left[l], right[l], js, root_res != nullptr, st[l], shared_nulls);
                for (Exec::Side* s : {&st[l].ls, &st[l].rs}) {
                    if (s->carry_mode == CARRY_ROWIDX) ok = false;  // a row index means nothing on another rank
                    if (s->carry_mode == CARRY_COLUMN && s->rel->cols[s->carry_col].kind == COL_IOTA) ok = false;
                    if (s->carry_mode == CARRY_WIDE)
                        for (int c : s->wide_cols)
                            if (s->rel->cols[c].kind == COL_IOTA || s->rel->cols[c].type == RJ_VARCHAR) ok = false;
                    ex_[l]->prepare_wide(*s);
                }
            });
            mine[l] = {left[l].n, right[l].n, ok ? 1ull : 0ull, (uint64_t)lerr[l].code};
        }
        gather(mine, 4, all);
        agree(all, 3, lerr, "preparing the join");
        uint64_t tot_left = 0, tot_right = 0;
        bool     ok = true;
        for (int r = 0; r < world_; ++r) {
            tot_left += all[r][0];
            tot_right += all[r][1];
            ok = ok && all[r][2] != 0;
        }
        auto all_empty = [&] {
            std::vector<Rel> out((size_t)nl_);
            for (int l = 0; l < nl_; ++l) {
                use(l);
                out[l] = ex_[l]->empty_rel(js, root_res ? (*root_res)[l] : nullptr);
            }
            return out;
        };
        // with an empty child the reference returns {} before looking at anything
        // (src/execute.cpp:50); a probe key of another type matches nothing (:65-71)
        if (tot_left == 0 || tot_right == 0 || st[0].type_mismatch) return all_empty();
        if (!ok)
            throw_fmt(RJ_ERR_UNSUPPORTED,
                      "sharded join: a side needs more payload than travels with the key (more than %d carry "
                      "words incl. a validity word, or a VARCHAR column): its row index would travel, and "
                      "means nothing on another rank",
                      MAX_WORDS - st[0].KW);
        const int KW = st[0].KW;

        // ---- the radix plan, from totals every rank knows: L local bits per rank, of which the
        // first digit (s bits) rides on stage A's pass together with the owner digit (rb_ bits) —
        // stage A fans out world * 2^s ways, and stage B starts at the second local digit
        const uint64_t tot_build = js.build_left ? tot_left : tot_right;
        const uint32_t Lbits = ex_[0]->join_bits(js, std::max<uint64_t>((tot_build + world_ - 1) / world_, 1), rb_);
        const uint32_t sbits = (g_->tune.fold_owner && Lbits >= 2 && rb_ < (uint32_t)PT_MAXBITS)
                                   ? std::min<uint32_t>(Lbits - 1, (uint32_t)PT_MAXBITS - rb_)
                                   : 0u;
        const uint32_t S = 1u << sbits, FA = (uint32_t)world_ * S;

        // ---- stage A on every local rank, both sides
        struct SideX {
            Parted                A;       // partitioned by (owner rank, first local digit)
            BufP                  recv[MAX_WORDS];
            WordSrc               ws;      // what arrived
            Ev                    ready, done, counted;
            ExchangePlan          plan;
            BufP                  segs;    // device copy of plan.seg_begin | seg_end | part_off
            Parted                P;       // stage B partitions
        };
        std::vector<SideX> bx((size_t)nl_), px((size_t)nl_);
        if (((size_t)FA + 1) * 4 > Context::SMALL_PINNED / 4)
            throw_fmt(RJ_ERR_UNSUPPORTED, "sharded join: more than %zu stage-A partitions", Context::SMALL_PINNED / 16 - 1);
        for (int l = 0; l < nl_; ++l) {
            guarded(lerr[l], [&] {
                use(l);
                inject_failure(2, l);
                Exec&    E = *ex_[l];
                Context* c = g_->lane(l);
                for (int side = 0; side < 2; ++side) {
                    Exec::Side& s = side == 0 ? st[l].bs() : st[l].ps();
                    SideX&      X = side == 0 ? bx[l] : px[l];
                    TupleSrc    src = E.make_src(st[l], s, js);
                    X.ready.make();
                    X.done.make();
                    X.counted.make();
                    // the per-partition offsets leave for the host BEFORE the scatter is enqueued: the
                    // host sizes and starts the exchange of the build side while the probe side's
                    // scatter is still running
                    // (pinned block: [0, 1/4) and [1/4, 1/2) = the two sides' stage-A offsets, [1/2, 3/4) and
                    // [3/4, 1) = their exchanged-run tables, below)
                    uint32_t* host_off = static_cast<uint32_t*>(c->small_pinned()) + (size_t)side * (Context::SMALL_PINNED / 16);
                    const std::function<void(const uint32_t*)> counts_out = [&](const uint32_t* off) {
                        RJ_HIP(hipMemcpyAsync(host_off, off, ((size_t)FA + 1) * 4, hipMemcpyDeviceToHost, c->stream));
                        RJ_HIP(hipEventRecord(X.counted.e, c->stream));
                    };
                    PassShape shape;
                    if (rb_ && sbits) {
                        shape.hi_shift = 32 - rb_;
                        shape.lo_bits = sbits;
                    }
                    // one field when only one of the two digits exists: the owner's (top bits) or the local one's (low bits)
                    const uint32_t shiftA = sbits ? 0u : (rb_ ? 32 - rb_ : 31);
                    X.A = E.partition(&src, nullptr, KW, s.CW, rb_ + sbits, shiftA, /*single pass*/ true, nullptr, &counts_out,
                                      shape.hi_shift ? &shape : nullptr);
                    // (test hook: this rank's probe-side slices are "never" ready — 7 s — so that the
                    // bounded wait for their exchange can be seen to expire; the counts left earlier)
                    if (g_->tune.debug_shard_fail == 4 && g_->tune.debug_shard_fail_rank == rank_base_ + l && side == 1)
                        launch_debug_stall(E.L, 7000);
                    RJ_HIP(hipEventRecord(X.ready.e, c->stream));
                }
            });
        }
        // ---- who holds how much for whom: cnt[src rank][side][dst rank][digit], + a status word
        const size_t per_side = (size_t)world_ * S;
        for (int l = 0; l < nl_; ++l) {
            mine[l].assign(2 * per_side + 1, 0);
            guarded(lerr[l], [&] {
                use(l);
                Context* c = g_->lane(l);
                for (int side = 0; side < 2; ++side) {
                    SideX& X = side == 0 ? bx[l] : px[l];
                    RJ_HIP(hipEventSynchronize(X.counted.e));  // (this rank's own stream: finite)
                    const uint32_t* off = static_cast<const uint32_t*>(c->small_pinned()) + (size_t)side * (Context::SMALL_PINNED / 16);
                    for (size_t q = 0; q < per_side; ++q) mine[l][side * per_side + q] = off[q + 1] - off[q];
                }
            });
            if (lerr[l].code) mine[l].assign(2 * per_side + 1, 0);
            mine[l][2 * per_side] = (uint64_t)lerr[l].code;
        }
        lap("stage A (histograms on the host; scatters may still run)");
        gather(mine, 2 * per_side + 1, all);
        agree(all, 2 * per_side, lerr, "partitioning its shard (stage A)");
        lap("count all-gather");

        // ---- the exchange layout (host arithmetic, rj_xplan.cpp).  Whether a rank would receive
        // more than 2^32 tuples is decided for EVERY rank of the world from the same tensor, so
        // all ranks throw the same error here — none of them enters the collective alone.
        std::vector<uint64_t> cnt[2];
        for (int side = 0; side < 2; ++side) {
            cnt[side].resize((size_t)world_ * per_side);
            for (int r = 0; r < world_; ++r)
                std::copy(all[r].begin() + side * per_side, all[r].begin() + (side + 1) * per_side,
                          cnt[side].begin() + (size_t)r * per_side);
            const int over = exchange_first_overflow((uint32_t)world_, S, cnt[side].data());
            if (over >= 0)
                throw_fmt(RJ_ERR_UNSUPPORTED, "sharded join: rank %d would receive more than 2^32 tuples of the %s side", over,
                          side == 0 ? "build" : "probe");
        }
        // receive buffers (a local allocation may fail: agreed on with one more status word)
        for (int l = 0; l < nl_; ++l) {
            guarded(lerr[l], [&] {
                use(l);
                inject_failure(3, l);
                Context* c = g_->lane(l);
                for (int side = 0; side < 2; ++side) {
                    SideX&    X = side == 0 ? bx[l] : px[l];
                    const int CW = side == 0 ? st[l].bs().CW : st[l].ps().CW;
                    exchange_plan((uint32_t)world_, S, (uint32_t)(rank_base_ + l), cnt[side].data(), X.plan);
                    X.ws.n = X.plan.n_recv;
                    X.ws.packed = X.A.packed;
                    for (int a = 0; a < MAX_WORDS; ++a) {
                        const uint32_t wbytes = array_width(X.A, KW, CW, a);
                        if (!wbytes) continue;
                        X.recv[a] = c->buf(std::max<uint64_t>(X.plan.n_recv, 1) * wbytes);
                        X.ws.w.w[a] = X.recv[a]->as<uint32_t>();
                    }
                    if (S > 1) {  // the runs that arrive, as input segments of stage B's first pass
                        // (through PINNED memory: an asynchronous copy from pageable memory makes the
                        // host wait for everything queued on the stream — stage A's scatters — and the
                        // exchange, which is to overlap them, would not even be enqueued until then)
                        const size_t ns = X.plan.seg_begin.size(), words = 2 * ns + S + 1;
                        if (words * 4 > Context::SMALL_PINNED / 4) throw_fmt(RJ_ERR_UNSUPPORTED, "sharded join: too many exchanged runs");
                        uint32_t* h = static_cast<uint32_t*>(c->small_pinned()) + Context::SMALL_PINNED / 8 + (size_t)side * (Context::SMALL_PINNED / 16);
                        std::copy(X.plan.seg_begin.begin(), X.plan.seg_begin.end(), h);
                        std::copy(X.plan.seg_end.begin(), X.plan.seg_end.end(), h + ns);
                        std::copy(X.plan.part_off.begin(), X.plan.part_off.end(), h + 2 * ns);
                        X.segs = c->buf(words * 4);
                        RJ_HIP(hipMemcpyAsync(X.segs->p, h, words * 4, hipMemcpyHostToDevice, c->stream));
                    }
                }
            });
            mine[l] = {(uint64_t)lerr[l].code};
        }
        gather(mine, 1, all);
        agree(all, 0, lerr, "allocating its receive buffers");

        // ---- the exchange: build side first, probe side queued behind it on the exchange streams
        for (int side = 0; side < 2; ++side) {
            std::vector<SideX>& XS = side == 0 ? bx : px;
            // arrays of the partition layout (the same on every rank: KW, CW and the packing rule agree)
            const int CW = side == 0 ? st[0].bs().CW : st[0].ps().CW;
            // one all-to-all per relation: every array of the layout travels in the same group
            std::vector<std::vector<XferSpec>> specs((size_t)nl_);
            std::vector<hipEvent_t>            ready, done;
            for (int l = 0; l < nl_; ++l) {
                SideX& X = XS[l];
                for (int a = 0; a < MAX_WORDS; ++a) {
                    const uint32_t wbytes = array_width(X.A, KW, CW, a);
                    if (!wbytes) continue;
                    XferSpec sp;
                    sp.send = reinterpret_cast<const uint8_t*>(X.A.w.w[a]);
                    sp.recv = reinterpret_cast<uint8_t*>(X.ws.w.w[a]);
                    for (int r = 0; r < world_; ++r) {
                        sp.send_off.push_back(X.plan.send_off[r] * wbytes);
                        sp.send_cnt.push_back(X.plan.send_cnt[r] * wbytes);
                        sp.recv_off.push_back(X.plan.recv_off[r] * wbytes);
                        sp.recv_cnt.push_back(X.plan.recv_cnt[r] * wbytes);
                    }
                    specs[l].push_back(std::move(sp));
                }
                ready.push_back(X.ready.e);
                done.push_back(X.done.e);
            }
            if (comm_) {
                comm_->all_to_all(specs, ready, done, side == 0 ? "exchange_build" : "exchange_probe");
            } else {  // one rank, no transport: the slice is the whole
                use(0);
                for (const XferSpec& sp : specs[0])
                    RJ_HIP(hipMemcpyAsync(sp.recv, sp.send, sp.send_cnt[0], hipMemcpyDeviceToDevice, g_->stream));
                RJ_HIP(hipEventRecord(done[0], g_->stream));
            }
        }

        lap("exchange (both sides)");
        // ---- stage B: each rank joins what it owns.  The host waits for an exchange with a
        // bound (Comm::wait_event) BEFORE it makes the compute stream depend on it, so no stream
        // and no host thread of this rank ever sits behind a peer that is gone.
        std::vector<Rel> out((size_t)nl_);
        auto stage_b = [&](int l, SideX& X, int CW) {
            Exec& E = *ex_[l];
            if (S == 1) return E.partition(nullptr, &X.ws, KW, CW, Lbits);
            const size_t ns = X.plan.seg_begin.size();
            PassShape    shape;
            shape.seg_begin = X.segs->as<uint32_t>();
            shape.seg_end = shape.seg_begin + ns;
            shape.oseg_off = shape.seg_begin + 2 * ns;
            shape.nseg = (uint32_t)ns;
            shape.n_oseg = S;
            shape.oseg_shift = rb_;  // `world` runs per digit
            shape.prior_bits = {sbits};
            return E.partition(nullptr, &X.ws, KW, CW, Lbits - sbits, sbits, false, nullptr, nullptr, &shape);
        };
        // (what fails on ONE rank from here on — an allocation, a result beyond 2^32 rows — is kept and
        // reported with the next join's first status word, or at the end of the plan: the peers are
        // not in a collective with this rank any more, but would be in the next join)
        if (pending_.empty()) pending_.resize((size_t)nl_);
        for (int side = 0; side < 2; ++side)
            for (int l = 0; l < nl_; ++l) {
                SideX& X = side == 0 ? bx[l] : px[l];
                if (comm_) comm_->wait_event(l, X.done.e, side == 0 ? "the exchange of the build side" : "the exchange of the probe side");
                guarded(pending_[l], [&] {
                    use(l);
                    RJ_HIP(hipStreamWaitEvent(g_->lane(l)->stream, X.done.e, 0));
                    X.P = stage_b(l, X, side == 0 ? st[l].bs().CW : st[l].ps().CW);
                });
            }
        for (int l = 0; l < nl_; ++l)
            guarded(pending_[l], [&] {
                use(l);
                inject_failure(6, l);
                st[l].cap_hint = std::max(bx[l].ws.n, px[l].ws.n);
                out[l] = ex_[l]->join_finish(st[l], js, &bx[l].P, &px[l].P, Lbits, root_res ? (*root_res)[l] : nullptr);
            });
        lap("stage B (passes + join)");
        // stage A's arrays were read by the exchange streams (and by peers): they may go back
        // to the block caches only now that every rank's outgoing copies have left
        sync_all();
        for (int l = 0; l < nl_; ++l) wait_xfer(l, "the exchange streams");
        return out;
    }

    void gather(const std::vector<std::vector<uint64_t>>& mine, size_t k,
                std::vector<std::vector<uint64_t>>& all) {
        if (comm_) {
            comm_->allgather_u64(mine, k, all);
        } else {
            all.assign(1, mine[0]);
        }
    }
};

bool node_shardable(const rj_plan* plan, uint64_t idx, int depth, std::string* why) {
    if (idx >= plan->n_nodes || depth > 4096) return false;
    const rj_node& n = plan->nodes[idx];
    if (n.kind == RJ_NODE_SCAN) {
        for (uint64_t k = 0; k < n.n_out; ++k)
            if (n.out_type[k] == RJ_VARCHAR) {
                if (why) *why = "a scan outputs a VARCHAR column";
                return false;
            }
        return true;
    }
    if (n.kind != RJ_NODE_JOIN) return false;
    if (!node_shardable(plan, n.left, depth + 1, why) || !node_shardable(plan, n.right, depth + 1, why))
        return false;
    if (n.left >= plan->n_nodes || n.right >= plan->n_nodes) return false;
    const uint64_t lw = plan->nodes[n.left].n_out, rw = plan->nodes[n.right].n_out;
    if (n.left_attr >= lw || n.right_attr >= rw) return false;
    // what must travel with the key on each side: up to MAX_WORDS - KW carry words (a nullable
    // column needs one word more for its validity bits — known only at run time, where the ranks
    // then agree to refuse the join together)
    const rj_node& build = n.build_left ? plan->nodes[n.left] : plan->nodes[n.right];
    const int32_t  key_type = build.out_type[n.build_left ? n.left_attr : n.right_attr];
    const int      KW = key_type == RJ_INT32 ? 1 : 2;
    std::set<uint64_t> need_l, need_r;
    for (uint64_t k = 0; k < n.n_out; ++k) {
        const uint64_t c = n.out_idx[k];
        if (c >= lw + rw) return false;
        if (c < lw) {
            if (c != n.left_attr) need_l.insert(c);
        } else if (c - lw != n.right_attr) {
            need_r.insert(c - lw);
        }
    }
    auto words = [&](const rj_node& child, const std::set<uint64_t>& need, int& n64) {
        int w = 0;
        n64 = 0;
        for (uint64_t c : need) {
            const bool wide = child.out_type[c] == RJ_INT64 || child.out_type[c] == RJ_FP64;
            w += wide ? 2 : 1;
            n64 += wide;
        }
        return w;
    };
    for (int side = 0; side < 2; ++side) {
        int       n64 = 0;
        const int w = words(side == 0 ? plan->nodes[n.left] : plan->nodes[n.right], side == 0 ? need_l : need_r, n64);
        if (w > MAX_WORDS - KW || n64 > 1 || (n64 == 1 && w == 4)) {
            if (why) *why = "a join side needs more payload than travels with the key (more than 3 carry words, or two 64-bit columns)";
            return false;
        }
    }
    return true;
}
}  // namespace

Result* execute_plan(Context* ctx, const rj_plan* plan, Table* const* tables, uint64_t n_tables,
                     int flags, TableFetch* fetch) {
    Exec e(ctx, plan, tables, n_tables, flags);
    e.set_fetch(fetch);
    return e.run();
}

Result* join_tuples(Context* ctx, const rj_tuples* build, const rj_tuples* probe,
                    uint32_t skip_rank_bits, int flags) {
    Exec e(ctx, nullptr, nullptr, 0, flags);
    return e.run_tuples(build, probe, skip_rank_bits);
}

void execute_sharded(Context* group, const rj_plan* plan, Table* const* tables, uint64_t n_inputs,
                     int flags, Result** out) {
    ShardedExec e(group, plan, tables, n_inputs, flags);
    e.run(out);
}

bool plan_shardable(const rj_plan* plan, std::string* why) {
    if (!plan || plan->root >= plan->n_nodes) return false;
    return node_shardable(plan, plan->root, 0, why);
}

void shard_partition(Context* ctx, const Table* t, uint64_t key_col, uint64_t carry_col,
                     uint32_t n_ranks, rj_tuples* out, uint64_t* counts) {
    Exec e(ctx, nullptr, nullptr, 0, 0);
    e.run_shard(t, key_col, carry_col, n_ranks, out, counts);
}

}  // namespace rj
