// rj_table.hip — device-resident ColumnarTables and result transfer.
//
// Replaces the page-walking half of Table::from_columnar (reference
// src/build_table.cpp:312-436): the reference's inputs are individually `new`-ed
// 8 KiB host blocks (include/plan.h:64-68), neither pinned nor contiguous, so
// they are gathered into pinned staging and copied to HBM in 32 MiB chunks.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "rj_internal.hpp"

namespace rj {

static constexpr size_t CHUNK_PAGES = 4096;  // 32 MiB per staging half

// Header scan of one fixed-width column that already sits in HBM.
static void analyse_column(Context* ctx, TableColumn& c, uint64_t num_rows) {
    c.regular = false;
    c.page_rows_total = 0;
    if (c.n_pages == 0) return;
    if (c.n_pages > 0xffffffffull) throw_fmt(RJ_ERR_UNSUPPORTED, "column has too many pages");
    const uint32_t rows_full = c.type == RJ_INT32 ? ROWS32 : ROWS64;
    c.page_rows = ctx->buf(c.n_pages * 4);
    BufP flags = ctx->buf(16);
    RJ_HIP(hipMemsetAsync(flags->p, 0, 16, ctx->stream));
    Launch L = ctx->launch();
    launch_page_headers(L, c.dev_pages, (uint32_t)c.n_pages, rows_full, c.page_rows->as<uint32_t>(),
                        flags->as<unsigned long long>());
    unsigned long long h[2] = {0, 0};
    RJ_HIP(hipMemcpyAsync(h, flags->p, 16, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    c.page_rows_total = h[1];
    c.regular = (h[0] == 0) && (h[1] == num_rows);
    if (h[1] > 0xfffffff0ull) throw_fmt(RJ_ERR_UNSUPPORTED, "column pages hold more than 2^32 rows");
    if (h[1] > num_rows) {
        // The pages hold more rows than the table declares.  The reference throws
        // std::runtime_error("row_idx") only when a NON-NULL value lands at a row index
        // >= num_rows (src/build_table.cpp:334-336); NULL rows past the end are tolerated.
        BufP row_base = ctx->buf((c.n_pages + 1) * 4);
        launch_scan_bins(L, c.page_rows->as<uint32_t>(), (uint32_t)c.n_pages,
                         row_base->as<uint32_t>(), nullptr);
        RJ_HIP(hipMemsetAsync(flags->p, 0, 8, ctx->stream));
        launch_rows_beyond(L, c.dev_pages, (uint32_t)c.n_pages, row_base->as<uint32_t>(), num_rows,
                           flags->as<unsigned long long>());
        RJ_HIP(hipMemcpyAsync(h, flags->p, 8, hipMemcpyDeviceToHost, ctx->stream));
        ctx->sync();
        if (h[0] != 0) throw_fmt(RJ_ERR_DATA, "row_idx");
    }
}

Table* table_adopt(Context* ctx, uint64_t num_rows, uint64_t n_cols, const int32_t* types,
                   const void* const* dev_pages, const uint64_t* n_pages) {
    if (num_rows > 0xfffffff0ull) throw_fmt(RJ_ERR_UNSUPPORTED, "more than 2^32 rows in one table");
    std::unique_ptr<Table> t(new rj_table());
    t->ctx = ctx;
    t->num_rows = num_rows;
    t->cols.resize(n_cols);
    for (uint64_t i = 0; i < n_cols; ++i) {
        TableColumn& c = t->cols[i];
        c.type = types[i];
        if (c.type < RJ_INT32 || c.type > RJ_VARCHAR) throw_fmt(RJ_ERR_ARG, "bad column type");
        c.n_pages = n_pages[i];
        if (c.type == RJ_VARCHAR) {
            if (dev_pages[i] != nullptr)
                throw_fmt(RJ_ERR_ARG, "VARCHAR columns cannot be adopted from device memory");
            c.n_pages = 0;
            continue;
        }
        c.dev_pages = static_cast<const uint8_t*>(dev_pages[i]);
        if (c.n_pages && !c.dev_pages) throw_fmt(RJ_ERR_ARG, "null page pointer");
        analyse_column(ctx, c, num_rows);
    }
    return t.release();
}

// ---- upload = prepare (caller's thread: shape the Table, reserve HBM) + fill (any thread:
//      gather into pinned staging, H2D in 32 MiB chunks, page headers read on the way)
namespace {

// every validity bit of the page's nr rows set?  (bitmap = last (nr+7)/8 bytes)
inline bool bitmap_all_ones(const uint8_t* page, uint32_t nr) {
    const uint32_t nb = (nr + 7) / 8;
    const uint8_t* bm = page + PAGE_BYTES - nb;
    const uint32_t full = nr / 8;
    uint32_t       k = 0;
    for (; k + 8 <= full; k += 8) {
        uint64_t w;
        memcpy(&w, bm + k, 8);
        if (w != ~0ull) return false;
    }
    for (; k < full; ++k)
        if (bm[k] != 0xff) return false;
    if (nr & 7u) {
        const uint8_t want = (uint8_t)((1u << (nr & 7u)) - 1u);
        if ((bm[full] & want) != want) return false;
    }
    return true;
}

struct UploadLane {
    hipStream_t stream = nullptr;
    uint8_t*    stage = nullptr;  // 2 * CHUNK_PAGES pages of pinned memory
    hipEvent_t  ev[2] = {nullptr, nullptr};
    bool        used[2] = {false, false};
    int         half = 0;
    UploadLane(hipStream_t s, void* pinned) : stream(s), stage(static_cast<uint8_t*>(pinned)) {
        RJ_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
        if (hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) {
            (void)hipEventDestroy(ev[0]);
            throw Error(RJ_ERR_DEVICE, "hipEventCreate failed");
        }
    }
    ~UploadLane() {
        (void)hipStreamSynchronize(stream);  // nothing may still read the staging buffer
        (void)hipEventDestroy(ev[0]);
        (void)hipEventDestroy(ev[1]);
    }
    UploadLane(const UploadLane&) = delete;
    UploadLane& operator=(const UploadLane&) = delete;
};

Table* table_prepare(Context* ctx, const rj_input* in, const std::vector<bool>* col_used,
                     bool borrow_varchar) {
    if (!in) throw_fmt(RJ_ERR_ARG, "null input");
    if (in->num_rows > 0xfffffff0ull)
        throw_fmt(RJ_ERR_UNSUPPORTED, "more than 2^32 rows in one table");
    std::unique_ptr<Table> t(new rj_table());
    t->ctx = ctx;
    t->num_rows = in->num_rows;
    t->cols.resize(in->n_cols);
    for (uint64_t ci = 0; ci < in->n_cols; ++ci) {
        const rj_column& hc = in->cols[ci];
        TableColumn&     c = t->cols[ci];
        c.type = hc.type;
        c.n_pages = hc.n_pages;
        if (c.type < RJ_INT32 || c.type > RJ_VARCHAR) throw_fmt(RJ_ERR_ARG, "bad column type");
        if (col_used && !(*col_used)[ci]) {  // no ScanNode outputs this column
            c.skipped = true;
            c.n_pages = 0;
            continue;
        }
        if (hc.n_pages && !hc.pages) throw_fmt(RJ_ERR_ARG, "null page pointer");
        if (c.type == RJ_VARCHAR) {
            c.vc_pages.resize(hc.n_pages);
            if (borrow_varchar)
                for (uint64_t p = 0; p < hc.n_pages; ++p)
                    c.vc_pages[p] = static_cast<const uint8_t*>(hc.pages[p]);
            else
                c.host_pages.resize(hc.n_pages * PAGE_BYTES);
            continue;
        }
        if (hc.n_pages == 0) continue;
        if (hc.n_pages > 0xffffffffull) throw_fmt(RJ_ERR_UNSUPPORTED, "column has too many pages");
        c.owned = ctx->buf(hc.n_pages * PAGE_BYTES);
        c.dev_pages = c.owned->as<uint8_t>();
        c.page_rows = ctx->buf(hc.n_pages * 4);
        c.page_rows_host.resize(hc.n_pages);
    }
    return t.release();
}

// The header walk of Table::from_columnar (reference src/build_table.cpp:326-336) rides on the
// gather: rows per page, whether the column is "regular", and the "row_idx" check.
void table_fill(Table* t, const rj_input* in, UploadLane& lane) {
    for (uint64_t ci = 0; ci < in->n_cols; ++ci) {
        const rj_column& hc = in->cols[ci];
        TableColumn&     c = t->cols[ci];
        if (c.skipped) continue;
        if (c.type == RJ_VARCHAR) {
            if (c.host_pages.empty()) continue;  // borrowed, or no pages
            uint8_t*           dst = c.host_pages.data();
            const void* const* pages = hc.pages;
            parallel_for(hc.n_pages, 256, [&](size_t b, size_t e) {
                for (size_t p = b; p < e; ++p) {
                    memcpy(dst + p * PAGE_BYTES, pages[p], PAGE_BYTES);
                    c.vc_pages[p] = dst + p * PAGE_BYTES;
                }
            });
            continue;
        }
        if (hc.n_pages == 0) continue;
        const uint32_t        rows_full = c.type == RJ_INT32 ? ROWS32 : ROWS64;
        std::atomic<uint64_t> irregular{0}, total{0};
        uint32_t*             prow = c.page_rows_host.data();
        for (uint64_t p0 = 0; p0 < hc.n_pages; p0 += CHUNK_PAGES) {
            uint64_t np = std::min<uint64_t>(CHUNK_PAGES, hc.n_pages - p0);
            uint8_t* s = lane.stage + (size_t)lane.half * CHUNK_PAGES * PAGE_BYTES;
            if (lane.used[lane.half]) RJ_HIP(hipEventSynchronize(lane.ev[lane.half]));
            const void* const* pages = hc.pages + p0;
            parallel_for(np, 256, [&](size_t b, size_t e) {
                uint64_t irr = 0, tot = 0;
                for (size_t p = b; p < e; ++p) {
                    copy_page(s + p * PAGE_BYTES, pages[p]);
                    const uint8_t* pg = static_cast<const uint8_t*>(pages[p]);
                    uint16_t       nr16;
                    memcpy(&nr16, pg, 2);
                    const uint32_t nr = nr16;
                    uint64_t gp = p0 + p;
                    prow[gp] = nr;
                    tot += nr;
                    // regular = full page + every validity bit set (the reference decodes from
                    // the bitmap alone, src/build_table.cpp:326-343; the header's non-null
                    // count is never read for fixed-width pages)
                    irr += !bitmap_all_ones(pg, nr) ||
                           (gp + 1 < hc.n_pages ? nr != rows_full : (nr > rows_full || nr == 0));
                }
                copy_pages_fence();
                irregular += irr;
                total += tot;
            });
            RJ_HIP(hipMemcpyAsync(c.owned->as<uint8_t>() + p0 * PAGE_BYTES, s, np * PAGE_BYTES,
                                  hipMemcpyHostToDevice, lane.stream));
            RJ_HIP(hipEventRecord(lane.ev[lane.half], lane.stream));
            lane.used[lane.half] = true;
            lane.half ^= 1;
        }
        c.page_rows_total = total.load();
        // more rows in the pages than the table declares: the reference throws
        // std::runtime_error("row_idx") when a NON-NULL value lands at a row index >= num_rows
        // (src/build_table.cpp:334-336); trailing NULL rows are tolerated.  Rare: serial walk.
        if (c.page_rows_total > t->num_rows) {
            uint64_t rb = 0;
            for (uint64_t p = 0; p < hc.n_pages; ++p) {
                const uint32_t nr = prow[p];
                if (rb + nr > t->num_rows) {
                    const uint8_t* pg = static_cast<const uint8_t*>(hc.pages[p]);
                    const uint8_t* bm = pg + PAGE_BYTES - (nr + 7) / 8;
                    for (uint32_t i = 0; i < nr; ++i)
                        if (rb + i >= t->num_rows && ((bm[i >> 3] >> (i & 7u)) & 1u))
                            throw_fmt(RJ_ERR_DATA, "row_idx");
                }
                rb += nr;
            }
        }
        c.regular = irregular.load() == 0 && c.page_rows_total == t->num_rows;
        if (!c.regular)  // K1 needs the rows per page on the device
            RJ_HIP(hipMemcpyAsync(c.page_rows->p, prow, hc.n_pages * 4, hipMemcpyHostToDevice,
                                  lane.stream));
    }
    RJ_HIP(hipStreamSynchronize(lane.stream));
}

}  // namespace

void upload_host_pages(Context* ctx, const uint8_t* const* pages, uint64_t n_pages, uint8_t* dev) {
    if (!n_pages) return;
    UploadLane lane(ctx->stream, ctx->staging(2 * CHUNK_PAGES * PAGE_BYTES));
    for (uint64_t p0 = 0; p0 < n_pages; p0 += CHUNK_PAGES) {
        const uint64_t np = std::min<uint64_t>(CHUNK_PAGES, n_pages - p0);
        uint8_t*       s = lane.stage + (size_t)lane.half * CHUNK_PAGES * PAGE_BYTES;
        if (lane.used[lane.half]) RJ_HIP(hipEventSynchronize(lane.ev[lane.half]));
        parallel_for(np, 256, [&](size_t b, size_t e) {
            for (size_t p = b; p < e; ++p) copy_page(s + p * PAGE_BYTES, pages[p0 + p]);
            copy_pages_fence();
        });
        RJ_HIP(hipMemcpyAsync(dev + p0 * PAGE_BYTES, s, np * PAGE_BYTES, hipMemcpyHostToDevice, lane.stream));
        RJ_HIP(hipEventRecord(lane.ev[lane.half], lane.stream));
        lane.used[lane.half] = true;
        lane.half ^= 1;
    }
    RJ_HIP(hipStreamSynchronize(lane.stream));
}

Table* table_upload(Context* ctx, const rj_input* in, const std::vector<bool>* col_used,
                    bool borrow_varchar) {
    std::unique_ptr<Table> t(table_prepare(ctx, in, col_used, borrow_varchar));
    UploadLane lane(ctx->stream, ctx->staging(2 * CHUNK_PAGES * PAGE_BYTES));
    table_fill(t.get(), in, lane);
    return t.release();
}

// ---------------------------------------------------------------- AsyncUpload --
struct AsyncUpload::Impl {
    Context*                            ctx = nullptr;
    const rj_plan*                      plan = nullptr;
    std::vector<std::unique_ptr<Table>> tables;
    std::vector<uint64_t>               order;  // used inputs, in the order the plan walk scans them
    std::vector<int>                    state;  // 0 = pending, 1 = ready, 2 = failed
    std::exception_ptr                  error;
    std::mutex                          m;
    std::condition_variable             cv;
    std::atomic<bool>                   cancel{false};
    std::thread                         th;
    hipEvent_t                          start = nullptr;
};

// scans in execution order: children left first (Exec::node)
static void scan_order(const rj_plan* plan, uint64_t idx, int depth, std::vector<bool>& seen,
                       std::vector<uint64_t>& order) {
    if (idx >= plan->n_nodes || depth > 4096) return;  // the plan walk reports it
    const rj_node& n = plan->nodes[idx];
    if (n.kind == RJ_NODE_SCAN) {
        if (n.base_table_id < seen.size() && !seen[n.base_table_id]) {
            seen[n.base_table_id] = true;
            order.push_back(n.base_table_id);
        }
    } else if (n.kind == RJ_NODE_JOIN) {
        scan_order(plan, n.left, depth + 1, seen, order);
        scan_order(plan, n.right, depth + 1, seen, order);
    }
}

AsyncUpload::AsyncUpload(Context* ctx, const rj_plan* plan, const std::vector<bool>& used,
                         const std::vector<std::vector<bool>>& col_used)
    : im_(new Impl()) {
    Impl& im = *im_;
    im.ctx = ctx;
    im.plan = plan;
    im.tables.resize(plan->n_inputs);
    im.state.assign(plan->n_inputs, 0);
    std::vector<bool> seen(plan->n_inputs, false);
    scan_order(plan, plan->root, 0, seen, im.order);
    rj_input none{};
    for (uint64_t i = 0; i < plan->n_inputs; ++i) {
        bool up = used[i] && seen[i];
        im.tables[i].reset(table_prepare(ctx, up ? &plan->inputs[i] : &none,
                                         up ? &col_used[i] : nullptr, /*borrow_varchar=*/true));
        if (!up) im.state[i] = 1;
    }
    if (im.order.empty()) return;
    // blocks handed out above may still be read by work queued on the context's stream
    hipStream_t up = ctx->upload_stream();
    void*       pinned = ctx->upload_staging(2 * CHUNK_PAGES * PAGE_BYTES);
    RJ_HIP(hipEventCreateWithFlags(&im.start, hipEventDisableTiming));
    RJ_HIP(hipEventRecord(im.start, ctx->stream));
    RJ_HIP(hipStreamWaitEvent(up, im.start, 0));
    const int device = ctx->device;
    im.th = std::thread([this, up, pinned, device] {
        Impl& im = *im_;
        auto  t0 = std::chrono::steady_clock::now();
        size_t k = 0;
        try {
            RJ_HIP(hipSetDevice(device));
            UploadLane lane(up, pinned);
            for (; k < im.order.size() && !im.cancel.load(); ++k) {
                uint64_t id = im.order[k];
                table_fill(im.tables[id].get(), &im.plan->inputs[id], lane);
                {
                    std::lock_guard<std::mutex> g(im.m);
                    im.state[id] = 1;
                }
                im.cv.notify_all();
            }
        } catch (...) {
            std::lock_guard<std::mutex> g(im.m);
            im.error = std::current_exception();
        }
        {
            std::lock_guard<std::mutex> g(im.m);
            for (; k < im.order.size(); ++k)
                if (im.state[im.order[k]] == 0) im.state[im.order[k]] = 2;
        }
        im.cv.notify_all();
        busy_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    });
}

Table* AsyncUpload::get(uint64_t id) {
    Impl& im = *im_;
    if (id >= im.tables.size()) throw_fmt(RJ_ERR_ARG, "scan: bad base_table_id");
    std::unique_lock<std::mutex> g(im.m);
    if (im.state[id] == 0) {
        auto t0 = std::chrono::steady_clock::now();
        im.cv.wait(g, [&] { return im.state[id] != 0; });
        wait_ms_ += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (im.state[id] == 2) {
        if (im.error) std::rethrow_exception(im.error);
        throw_fmt(RJ_ERR_DEVICE, "upload cancelled");
    }
    return im.tables[id].get();
}

AsyncUpload::~AsyncUpload() {
    Impl& im = *im_;
    im.cancel.store(true);
    if (im.th.joinable()) im.th.join();
    if (im.start) (void)hipEventDestroy(im.start);
}

// D2H of `n_pages` contiguous page images on `ctx`'s device into caller-owned 8 KiB blocks;
// the copy of chunk k+1 overlaps the host scatter of chunk k
static void copy_part_pages(Context* ctx, const uint8_t* dev, uint64_t n_pages, void* const* dst) {
    RJ_HIP(hipSetDevice(ctx->device));
    uint8_t*   stage = static_cast<uint8_t*>(ctx->staging(2 * CHUNK_PAGES * PAGE_BYTES));
    uint64_t   n_chunks = (n_pages + CHUNK_PAGES - 1) / CHUNK_PAGES;
    hipEvent_t ev[2];
    RJ_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    RJ_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    auto issue = [&](uint64_t k) {
        uint64_t p0 = k * CHUNK_PAGES, np = std::min<uint64_t>(CHUNK_PAGES, n_pages - p0);
        int      h = (int)(k & 1);
        RJ_HIP(hipMemcpyAsync(stage + (size_t)h * CHUNK_PAGES * PAGE_BYTES, dev + p0 * PAGE_BYTES,
                              np * PAGE_BYTES, hipMemcpyDeviceToHost, ctx->stream));
        RJ_HIP(hipEventRecord(ev[h], ctx->stream));
    };
    try {
        issue(0);
        for (uint64_t k = 0; k < n_chunks; ++k) {
            int h = (int)(k & 1);
            RJ_HIP(hipEventSynchronize(ev[h]));
            if (k + 1 < n_chunks) issue(k + 1);
            uint64_t       p0 = k * CHUNK_PAGES, np = std::min<uint64_t>(CHUNK_PAGES, n_pages - p0);
            const uint8_t* s = stage + (size_t)h * CHUNK_PAGES * PAGE_BYTES;
            void* const*   d = dst + p0;
            parallel_for(np, 256, [&](size_t b, size_t e) {
                // plain memcpy here: measured faster than streaming stores for pinned -> pageable
                for (size_t p = b; p < e; ++p) memcpy(d[p], s + p * PAGE_BYTES, PAGE_BYTES);
            });
        }
    } catch (...) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipEventDestroy(ev[0]);
        (void)hipEventDestroy(ev[1]);
        throw;
    }
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
}

// Result pages -> caller-owned 8 KiB blocks (e.g. `new Page`, so the harness's
// Column::~Column, reference include/plan.h:95-99, can delete them).
void result_copy_pages(Result* r, uint64_t col, void* const* dst, uint64_t n_dst) {
    if (col >= r->cols.size()) throw_fmt(RJ_ERR_ARG, "column out of range");
    ResultColumn& c = r->cols[col];
    if (n_dst < c.n_pages) throw_fmt(RJ_ERR_ARG, "destination has too few pages");
    if (c.n_pages == 0) return;
    if (!c.dev_pages && c.more.empty()) {  // host-encoded pages (small VARCHAR results)
        const uint8_t* src = c.host_pages.data();
        parallel_for(c.n_pages, 256, [&](size_t b, size_t e) {
            for (size_t p = b; p < e; ++p) memcpy(dst[p], src + p * PAGE_BYTES, PAGE_BYTES);
        });
        return;
    }
    // the column's pages may sit on several devices (a result gathered from a sharded run)
    std::vector<ResultColumn::Part> parts;
    parts.push_back({r->ctx, c.dev_pages, c.more.empty() ? c.n_pages : c.n_first});
    for (const ResultColumn::Part& p : c.more) parts.push_back(p);
    uint64_t done = 0;
    for (const ResultColumn::Part& part : parts) {
        if (part.n_pages == 0) continue;
        copy_part_pages(part.ctx, part.pages->as<uint8_t>(), part.n_pages, dst + done);
        done += part.n_pages;
    }
    RJ_HIP(hipSetDevice(r->ctx->device));
}

// ------------------------------------------------------ rj_execute over several devices
namespace {

// all pages but the last full, every validity bit set, rows add up: addressable in place
bool host_column_regular(const rj_column& hc, uint64_t num_rows) {
    if (hc.type != RJ_INT32 && hc.type != RJ_INT64 && hc.type != RJ_FP64) return false;
    if (hc.n_pages == 0) return num_rows == 0;
    const uint32_t        rows_full = hc.type == RJ_INT32 ? ROWS32 : ROWS64;
    std::atomic<uint64_t> bad{0}, total{0};
    parallel_for(hc.n_pages, 1024, [&](size_t b, size_t e) {
        uint64_t nb = 0, tot = 0;
        for (size_t p = b; p < e; ++p) {
            const uint8_t* pg = static_cast<const uint8_t*>(hc.pages[p]);
            uint16_t       nr;
            memcpy(&nr, pg, 2);
            tot += nr;
            nb += !bitmap_all_ones(pg, nr) || (p + 1 < hc.n_pages ? nr != rows_full : (nr > rows_full || nr == 0));
        }
        bad += nb;
        total += tot;
    });
    return bad.load() == 0 && total.load() == num_rows;
}

}  // namespace

Result* execute_host_sharded(Context* g, const rj_plan* plan, const std::vector<bool>& used,
                             const std::vector<std::vector<bool>>& col_used) {
    const int nl = g->n_lanes();
    if (nl < 2 || !g->comm || g->comm->world() != nl) return nullptr;
    if (!plan_shardable(plan, nullptr)) return nullptr;
    for (uint64_t i = 0; i < plan->n_inputs; ++i) {
        if (!used[i]) continue;
        const rj_input& in = plan->inputs[i];
        for (uint64_t c = 0; c < in.n_cols; ++c)
            if (col_used[i][c] && !host_column_regular(in.cols[c], in.num_rows)) return nullptr;
    }
    // Row cuts at multiples of 1984 * 1007 rows: a page boundary of INT32 and of INT64/FP64
    // columns alike (the two page capacities are coprime), so a shard is a sub-range of every
    // column's page pointers.
    constexpr uint64_t U = (uint64_t)ROWS32 * ROWS64;
    // inputs below nl cut units would land on the last rank alone (every other shard empty) and
    // still pay the uploads, the gathers and the exchange: such plans run on the first device
    uint64_t largest = 0;
    for (uint64_t i = 0; i < plan->n_inputs; ++i)
        if (used[i]) largest = std::max<uint64_t>(largest, plan->inputs[i].num_rows);
    if (largest < (uint64_t)nl * U) return nullptr;
    std::vector<std::vector<std::unique_ptr<Table>>> tabs((size_t)nl);
    std::vector<Table*>                              flat((size_t)nl * plan->n_inputs, nullptr);
    std::vector<std::vector<rj_column>>              views((size_t)nl * plan->n_inputs);  // the shards' column views
    std::vector<rj_input>                            shard((size_t)nl * plan->n_inputs, rj_input{});
    for (int l = 0; l < nl; ++l)
        for (uint64_t i = 0; i < plan->n_inputs; ++i) {
            if (!used[i]) continue;
            const rj_input& in = plan->inputs[i];
            const uint64_t  r0 = l == 0 ? 0 : (in.num_rows * (uint64_t)l / nl) / U * U;
            const uint64_t  r1 = l + 1 == nl ? in.num_rows : (in.num_rows * (uint64_t)(l + 1) / nl) / U * U;
            std::vector<rj_column>& vc = views[(size_t)l * plan->n_inputs + i];
            vc.assign(in.cols, in.cols + in.n_cols);
            for (uint64_t k = 0; k < in.n_cols; ++k) {
                if (!col_used[i][k]) continue;
                const uint64_t rf = vc[k].type == RJ_INT32 ? ROWS32 : ROWS64;
                const uint64_t p0 = r0 / rf, p1 = (r1 + rf - 1) / rf;
                vc[k].pages = in.cols[k].pages + p0;
                vc[k].n_pages = p1 - p0;
            }
            rj_input& view = shard[(size_t)l * plan->n_inputs + i];
            view.num_rows = r1 - r0;
            view.n_cols = in.n_cols;
            view.cols = vc.data();
        }
    // one uploading thread per device: every lane gathers pages into its own pinned staging and
    // keeps its own PCIe link busy (the gathers take turns on the host workers, the copies overlap)
    std::vector<std::exception_ptr> failed((size_t)nl);
    auto upload_lane = [&](int l) {
        try {
            Context* c = g->lane(l);
            RJ_HIP(hipSetDevice(c->device));
            for (uint64_t i = 0; i < plan->n_inputs; ++i) {
                tabs[l].emplace_back(table_upload(c, &shard[(size_t)l * plan->n_inputs + i], used[i] ? &col_used[i] : nullptr, true));
                flat[(size_t)l * plan->n_inputs + i] = tabs[l].back().get();
            }
        } catch (...) {
            failed[l] = std::current_exception();
        }
    };
    {
        std::vector<std::thread> th;
        for (int l = 1; l < nl; ++l) th.emplace_back(upload_lane, l);
        upload_lane(0);
        for (std::thread& t : th) t.join();
    }
    for (int l = 0; l < nl; ++l)
        if (failed[l]) std::rethrow_exception(failed[l]);
    std::vector<Result*> parts((size_t)nl, nullptr);
    execute_sharded(g, plan, flat.data(), plan->n_inputs, 0, parts.data());
    // gather: rank 0's result takes the others' pages behind its own (pages of a Column need
    // not be full, reference src/build_table.cpp:326-343, so concatenation is a valid column)
    std::unique_ptr<Result> res(parts[0]);
    for (ResultColumn& rc : res->cols) rc.n_first = rc.n_pages;
    for (int l = 1; l < nl; ++l) {
        std::unique_ptr<Result> p(parts[l]);
        res->num_rows += p->num_rows;
        for (size_t k = 0; k < res->cols.size() && k < p->cols.size(); ++k) {
            ResultColumn& rc = res->cols[k];
            if (p->cols[k].n_pages == 0) continue;
            rc.more.push_back({p->ctx, p->cols[k].dev_pages, p->cols[k].n_pages});
            rc.n_pages += p->cols[k].n_pages;
        }
    }
    RJ_HIP(hipSetDevice(g->device));
    return res.release();
}

}  // namespace rj
