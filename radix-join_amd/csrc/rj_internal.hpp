// rj_internal.hpp — host-side runtime of librj.so: context, HBM block cache,
// per-kernel HIP-event profiler, resident tables and results.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rj.h"
#include "rj_comm.hpp"
#include "rj_device.hpp"
#include "rj_kernels.hpp"

namespace rj {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void throw_fmt(int code, const char* fmt, ...) {
    char    buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

#define RJ_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            ::rj::throw_fmt(_e == hipErrorOutOfMemory ? RJ_ERR_NOMEM : RJ_ERR_DEVICE,         \
                            "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                            __LINE__);                                                        \
    } while (0)

// ---------------------------------------------------------------- HBM cache --
// hipMalloc/hipFree of multi-GB buffers cost milliseconds and synchronise; the
// context keeps freed blocks and hands them out again.  All work of a context
// runs on ONE stream, so reuse is ordered by the stream.
class DevPool {
   public:
    void* alloc(size_t bytes);
    void  release(void* p);
    void  trim();  // hipFree everything that is not in use
    ~DevPool();
    size_t bytes_in_use() const { return in_use_; }
    size_t bytes_cached() const { return cached_; }
    // diagnostics (RJ_DIAG=2): what went to the driver instead of the cache
    uint64_t n_malloc = 0, n_trim = 0;
    double   malloc_ms = 0;

   private:
    struct Block {
        void*  p;
        size_t size;
        bool   free;
    };
    std::vector<Block> blocks_;
    size_t             in_use_ = 0, cached_ = 0;
};

struct Context;

// RAII device buffer drawn from the context's cache.
struct Buf {
    Context* ctx = nullptr;
    void*    p = nullptr;
    size_t   bytes = 0;
    Buf(Context* c, size_t n);
    ~Buf();
    Buf(const Buf&) = delete;
    Buf& operator=(const Buf&) = delete;
    template <class T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
};
using BufP = std::shared_ptr<Buf>;

// ----------------------------------------------------------------- profiler --
class Profiler {
   public:
    bool        on = false;
    int         level = 1;  // 1: hot kernels only, 2: every launch
    hipStream_t stream = nullptr;
    // start/stop events for one launch, or false when this launch is not timed
    bool        timed(const char* name, hipEvent_t* start, hipEvent_t* stop);
    void        drain();  // synchronises, folds finished records into totals
    void        reset();
    struct Tot {
        uint64_t launches = 0;
        double   ms = 0;
    };
    std::map<std::string, Tot> totals;
    std::vector<std::string>   order;
    ~Profiler();

   private:
    struct Rec {
        const char* name;
        hipEvent_t  a, b;
    };
    std::vector<Rec>        open_;
    std::vector<hipEvent_t> spare_;
    hipEvent_t              get_event();
};

// ------------------------------------------------------------------- tables --
struct TableColumn {
    int32_t  type = 0;
    uint64_t n_pages = 0;
    // fixed-width columns: page images in HBM
    const uint8_t* dev_pages = nullptr;
    BufP           owned;          // set when the table owns the images (upload)
    bool           regular = false; // addressable in place (see ColKind::COL_PAGED)
    uint64_t       page_rows_total = 0;
    BufP           page_rows;      // u32[n_pages] rows per page (for K1)
    std::vector<uint32_t> page_rows_host;  // the same, read off the headers while uploading
    // VARCHAR columns stay on the host: vc_pages[i] points at page i — into `host_pages`
    // (a private copy, resident tables) or into the caller's pages (rj_execute, where the
    // Plan's inputs outlive the call)
    std::vector<uint8_t>        host_pages;
    std::vector<const uint8_t*> vc_pages;
    // VARCHAR pages + row directory in HBM, uploaded on first use by a large root output
    // (device-side materialisation, rj_varchar_dev.hip)
    mutable BufP vc_dev, vc_dev_dir;
    bool                        skipped = false;  // not referenced by the plan: never uploaded
};

struct Table {
    Context*                 ctx = nullptr;
    uint64_t                 num_rows = 0;
    std::vector<TableColumn> cols;
};

// ------------------------------------------------------------------ results --
struct ResultColumn {
    int32_t              type = 0;
    uint64_t             n_pages = 0;   // all pages of the column (dev_pages + more)
    BufP                 dev_pages;   // fixed-width
    std::vector<uint8_t> host_pages;  // VARCHAR (encoded on the host)
    // a result gathered from several devices (rj_execute on a multi-device context): the pages
    // of the other ranks, in rank order behind dev_pages (which then holds n_first pages)
    struct Part {
        Context* ctx;
        BufP     pages;
        uint64_t n_pages;
    };
    std::vector<Part> more;
    uint64_t          n_first = 0;  // pages in dev_pages when `more` is in use
};

struct Result {
    Context*                  ctx = nullptr;
    uint64_t                  num_rows = 0;
    std::vector<ResultColumn> cols;
};

// Tuning / diagnostic knobs, read from the environment ONCE when the context is created (the
// plan walk is a hot host path: no getenv per join).
struct Tuning {
    int blocked_mid = 1;  // RJ_TUNE_BLOCKED_MID: packed pairs between the passes in blocks of 256 keys + 256 carries, so that
                          // the next pass' histogram reads the keys only (4 instead of 8 bytes per tuple)
    int mall_chunk = 0;   // RJ_TUNE_MALL_CHUNK: segments per chunk of the SECOND pass of packed plans — histogram, scan and
                          // scatter launched chunk by chunk on two streams, so that the scatter re-reads what the
                          // histogram just read from the 256 MiB Infinity Cache instead of HBM (0 = one launch each)
    int radix_bits = 0;   // RJ_TUNE_RADIX_BITS: total radix bits of every partitioned join (experiments; rj_config.radix_bits wins)
    int p1_bits = 0;      // RJ_TUNE_P1_BITS: radix bits of pass 1 in a two-pass plan (0 = even split)
    int fine = 1;         // RJ_TUNE_FINE: fine (two-digit) histogram for plans <= 2^PT_FINEBITS partitions
    int pack = 1;         // RJ_TUNE_PACK: 0 never, 1 always, 2 fine plans only: {key, carry} pairs
    int aos3 = 1;         // RJ_TUNE_AOS3: last pass of key + two-word-carry plans writes 12-byte tuples
    int aos_mid = 2;      // RJ_TUNE_AOS_MID: ... and so do the passes before it (+ a 16-bit digit side array
                          // for the next histogram); 0: key array + pair array between the passes
    int tpg1 = 0;         // RJ_TUNE_TPG1: tiles per group of pass 1 (0 = auto)
    int packed_side = 0;  // RJ_TUNE_PACKED_SIDE: digit side array between the passes of packed plans
    int tpg2 = 0;         // RJ_TUNE_TPG2: tiles per group of the later passes (0 = auto)
    int xcd_min_rows = 40 << 20;  // RJ_TUNE_XCD_MIN_ROWS: ... for passes over at least this many tuples
    int xcd_split = 1;    // RJ_TUNE_XCD_SPLIT: XCD-aware output placement of the big passes (PassParams::xcd_log2, xcd_remap)
    int bcast = 1;        // RJ_TUNE_BCAST: broadcast join for build sides that fit one LDS table
    int diag = 0;         // RJ_DIAG: 1 = join phase stamps, 2 = host-side timings on stderr
    int varchar_dev_rows = 200000;  // RJ_TUNE_VARCHAR_DEV: root VARCHAR columns of at least this many
                                    // rows are gathered + encoded on the device (0 = never)
    int vkey_hash_bits = 0;  // RJ_DEBUG_VKEY_HASH_BITS: keep only this many bits of a VARCHAR key's hash
                             // (tests: forces collisions through the verify + compact path)
    int sync_upload = 0;  // RJ_SYNC_UPLOAD: upload every input before the plan starts
    int wide_carry = 1;   // RJ_TUNE_WIDE_CARRY: several payload columns of a side (and NULL-bearing ones) travel
                          // with the key when they fit MAX_WORDS - KW carry words; 0: a row index travels and
                          // every column is gathered afterwards (k_gather)
    int fold_owner = 1;   // RJ_TUNE_FOLD_OWNER: a sharded join's stage A partitions by (owner rank, first local digit)
                          // in one pass; 0: by owner rank only (stage B then runs one more pass)
    int exchange_timeout_ms = 120000;  // RJ_EXCHANGE_TIMEOUT_MS: bound on every wait of the exchange step of a
                                       // sharded join (communicator bring-up, count gathers, the all-to-all)
    int bringup_timeout_ms = 0;        // RJ_BRINGUP_TIMEOUT_MS: its own bound for the bring-up (0: the same) — in a
                                       // cold process RCCL takes seconds to load, a running exchange milliseconds
    // RJ_DEBUG_SHARD_FAIL (tests): global rank RJ_DEBUG_SHARD_FAIL_RANK fails locally at this point of a
    // sharded join — 1: while preparing, 2: in stage A, 3: allocating its receive buffers, 5: in a scan below it, 6: building / probing what arrived; 4: it does not
    // fail but stalls (its probe-side slices are not ready for 7 s: the exchange's bounded wait expires)
    int debug_shard_fail = 0, debug_shard_fail_rank = 0;
    void from_env();
};

// ------------------------------------------------------------------ context --
struct Context {
    Tuning      tune;
    int         device = 0;
    hipStream_t stream = nullptr;
    bool        own_stream = false;
    int         radix_bits_override = 0;
    DevPool     pool;
    Profiler    prof;
    std::string last_error;
    // pinned staging for H2D / D2H of page images
    void*  pinned = nullptr;
    size_t pinned_bytes = 0;
    static constexpr size_t SMALL_PINNED = 32768;
    void*  pinned_small = nullptr;  // SMALL_PINNED bytes for counters that travel to the host
    void*  small_pinned();
    // second lane (rj_execute): inputs are uploaded by a helper thread on their own stream
    // while the plan already runs on `stream`
    hipStream_t copy_stream = nullptr;
    void*       pinned_up = nullptr;
    size_t      pinned_up_bytes = 0;
    hipStream_t upload_stream();
    hipStream_t aux = nullptr;   // second compute stream (a later radix pass run chunk by chunk alternates)
    hipStream_t aux_stream();
    void*       upload_staging(size_t bytes);
    Launch launch() {
        Launch L;
        L.stream = stream;
        L.self = this;
        L.timed = prof.on ? [](void* s, const char* n, hipEvent_t* a, hipEvent_t* b) {
            return static_cast<Context*>(s)->prof.timed(n, a, b);
        } : (bool (*)(void*, const char*, hipEvent_t*, hipEvent_t*)) nullptr;
        return L;
    }
    // multi-GPU: a context created over several devices is lane 0 of a group and owns one
    // further single-device context per additional device ("lanes", one rank each) plus the
    // exchange transport; a lane points back at its group
    std::vector<Context*>  peers;  // lanes 1..n-1 (owned)
    std::unique_ptr<Comm>  comm;   // set on lane 0 when the job has more than one rank
    Context*               group = nullptr;
    int                    n_lanes() const { return 1 + (int)peers.size(); }
    Context*               lane(int i) { return i == 0 ? this : peers[(size_t)i - 1]; }
    int   n_cu = 0;
    int   compute_units();  // CUs of the device (persistent-kernel grids)
    void  prewarm();        // RJ_CTX_PREWARM: one-off set-up costs now instead of in the first rj_execute
    BufP  buf(size_t bytes) { return std::make_shared<Buf>(this, bytes ? bytes : 16); }
    void* staging(size_t bytes);
    void  sync() { RJ_HIP(hipStreamSynchronize(stream)); }
    ~Context();
};

// Where a ScanNode gets its base table from: a plain array (resident tables) or the
// asynchronous uploader, whose get() blocks until that table has landed in HBM.
struct TableFetch {
    virtual Table* get(uint64_t id) = 0;
    virtual ~TableFetch() = default;
};

// rj_exec.hip
// tables: n_tables resident tables, or nullptr when `fetch` delivers them
Result* execute_plan(Context* ctx, const rj_plan* plan, Table* const* tables, uint64_t n_tables,
                     int flags, TableFetch* fetch = nullptr);
Result* join_tuples(Context* ctx, const rj_tuples* build, const rj_tuples* probe,
                    uint32_t skip_rank_bits, int flags);
// Sharded execution over the lanes of `group` (collective across the job's processes):
// tables[l * n_inputs + i] = lane l's shard of input i; out[l] = lane l's slice of the result.
void    execute_sharded(Context* group, const rj_plan* plan, Table* const* tables, uint64_t n_inputs,
                        int flags, Result** out);
// Can every JoinNode of the plan run sharded (at most one fixed-width non-key column per side)?
bool    plan_shardable(const rj_plan* plan, std::string* why);
void    shard_partition(Context* ctx, const Table* t, uint64_t key_col, uint64_t carry_col,
                        uint32_t n_ranks, rj_tuples* out, uint64_t* counts);

// rj_table.hip
// col_used: upload only these columns (nullptr = all); borrow_varchar: keep pointers to the
// caller's VARCHAR pages instead of copying them
Table* table_upload(Context* ctx, const rj_input* host, const std::vector<bool>* col_used = nullptr,
                    bool borrow_varchar = false);
// All inputs of a Plan, uploaded by a helper thread in the order the plan walk scans them
// (rj_execute).  HBM is reserved up front on the caller's thread; the helper only gathers,
// copies and reads page headers, so the block cache stays single-threaded.
class AsyncUpload : public TableFetch {
   public:
    AsyncUpload(Context* ctx, const rj_plan* plan, const std::vector<bool>& used,
                const std::vector<std::vector<bool>>& col_used);
    ~AsyncUpload() override;
    Table* get(uint64_t id) override;
    double busy_ms() const { return busy_ms_; }   // helper thread: gather + H2D
    double wait_ms() const { return wait_ms_; }   // plan walk: blocked in get()

   private:
    struct Impl;
    std::unique_ptr<Impl> im_;
    double                busy_ms_ = 0, wait_ms_ = 0;
};
Table* table_adopt(Context* ctx, uint64_t num_rows, uint64_t n_cols, const int32_t* types,
                   const void* const* dev_pages, const uint64_t* n_pages);
void   result_copy_pages(Result* r, uint64_t col, void* const* dst, uint64_t n_dst);
// n_pages individually allocated host pages -> contiguous page images at `dev` (pinned staging,
// 32 MiB chunks, on the context's stream; synchronises)
void   upload_host_pages(Context* ctx, const uint8_t* const* pages, uint64_t n_pages, uint8_t* dev);
// rj_execute on a context that owns several devices: shard the host inputs by row ranges, run
// the plan sharded and gather the ranks' pages into one result.  nullptr = this plan / these
// inputs cannot be sharded (the caller runs them on the first device).
Result* execute_host_sharded(Context* group, const rj_plan* plan, const std::vector<bool>& used,
                             const std::vector<std::vector<bool>>& col_used);

// rj_ingest.hip — Table::from_csv on the device
Table*   table_from_csv(Context* ctx, const char* text, uint64_t n_bytes, uint64_t n_cols, const int32_t* col_type,
                        const rj_filter_op* filter, uint64_t n_filter_ops);
int parse_fp64_host(const char* s, uint64_t n, uint64_t* bits);  // rj_fp64.hpp on the host: 0 parsed, 1 out of range, 2 undecided
uint64_t table_col_pages(const Table* t, uint64_t col);
void     table_copy_pages(Context* ctx, const Table* t, uint64_t col, void* const* dst, uint64_t n_dst);

// rj_hostpool.cpp (host only)
// fn(b, e) over disjoint ranges of at most `grain` items covering [0, n), on the process-wide
// worker threads plus the caller; the first exception thrown by fn is rethrown here.
void parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t)>& fn);

// One 8 KiB page with streaming stores when the destination is 32-byte aligned (else memcpy);
// copy_pages_fence() orders a thread's streaming stores before it reports its range done.
void copy_page(void* dst, const void* src);
void copy_pages_fence();

// rj_varchar.cpp (host only)
// Page directory of a VARCHAR column: row_base[p] = rows before page p, row_base[n_pages] =
// rows the pages hold.  Throws "row_idx" if they hold more than num_rows.
void varchar_dir_build(const uint8_t* const* pages, uint64_t n_pages, uint64_t num_rows,
                       std::vector<uint64_t>& row_base);
// Gather the strings of rows idx[0..n) and encode them as VARCHAR pages (reference fill rule
// src/build_table.cpp:595-677).
void varchar_gather_encode(const uint8_t* const* pages, uint64_t n_pages,
                           const std::vector<uint64_t>& row_base, const uint32_t* idx, uint64_t n,
                           std::vector<uint8_t>& out_pages, uint64_t& n_out_pages);

}  // namespace rj

// Opaque C handles
struct rj_context : rj::Context {};
struct rj_table : rj::Table {};
struct rj_result : rj::Result {};
