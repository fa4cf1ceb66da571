// rj_table.hip — device-resident ColumnarTables and result transfer.
//
// Replaces the page-walking half of Table::from_columnar (reference
// src/build_table.cpp:312-436): the reference's inputs are individually `new`-ed
// 8 KiB host blocks (include/plan.h:64-68), neither pinned nor contiguous, so
// they are gathered into pinned staging and copied to HBM in 32 MiB chunks.
#include <algorithm>
#include <functional>
#include <thread>

#include "rj_internal.hpp"

namespace rj {

static constexpr size_t CHUNK_PAGES = 4096;  // 32 MiB per staging half

static void parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t)>& fn) {
    unsigned hw = std::thread::hardware_concurrency();
    size_t   nt = std::min<size_t>(hw ? hw : 4, 16);
    nt = std::min(nt, (n + grain - 1) / grain);
    if (nt <= 1) {
        fn(0, n);
        return;
    }
    std::vector<std::thread> th;
    size_t                   per = (n + nt - 1) / nt;
    for (size_t t = 0; t < nt; ++t) {
        size_t b = t * per, e = std::min(n, b + per);
        if (b >= e) break;
        th.emplace_back([=, &fn] { fn(b, e); });
    }
    for (auto& t : th) t.join();
}

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
    // more rows in the pages than the table declares: the reference throws
    // std::runtime_error("row_idx") (src/build_table.cpp:334-336)
    if (h[1] > num_rows) throw_fmt(RJ_ERR_DATA, "row_idx");
    c.regular = (h[0] == 0) && (h[1] == num_rows);
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

Table* table_upload(Context* ctx, const rj_input* in, const std::vector<bool>* col_used,
                    bool borrow_varchar) {
    if (!in) throw_fmt(RJ_ERR_ARG, "null input");
    if (in->num_rows > 0xfffffff0ull)
        throw_fmt(RJ_ERR_UNSUPPORTED, "more than 2^32 rows in one table");
    std::unique_ptr<Table> t(new rj_table());
    t->ctx = ctx;
    t->num_rows = in->num_rows;
    t->cols.resize(in->n_cols);
    uint8_t* stage = static_cast<uint8_t*>(ctx->staging(2 * CHUNK_PAGES * PAGE_BYTES));
    hipEvent_t ev[2];
    RJ_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    RJ_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    bool used[2] = {false, false};
    int  half = 0;
    try {
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
            if (c.type == RJ_VARCHAR) {
                const void* const* pages = hc.pages;
                c.vc_pages.resize(hc.n_pages);
                if (borrow_varchar) {
                    for (uint64_t p = 0; p < hc.n_pages; ++p)
                        c.vc_pages[p] = static_cast<const uint8_t*>(pages[p]);
                } else {
                    c.host_pages.resize(hc.n_pages * PAGE_BYTES);
                    uint8_t* dst = c.host_pages.data();
                    parallel_for(hc.n_pages, 256, [&](size_t b, size_t e) {
                        for (size_t p = b; p < e; ++p) {
                            memcpy(dst + p * PAGE_BYTES, pages[p], PAGE_BYTES);
                            c.vc_pages[p] = dst + p * PAGE_BYTES;
                        }
                    });
                }
                continue;
            }
            if (hc.n_pages == 0) continue;
            c.owned = ctx->buf(hc.n_pages * PAGE_BYTES);
            c.dev_pages = c.owned->as<uint8_t>();
            for (uint64_t p0 = 0; p0 < hc.n_pages; p0 += CHUNK_PAGES) {
                uint64_t np = std::min<uint64_t>(CHUNK_PAGES, hc.n_pages - p0);
                uint8_t* s = stage + (size_t)half * CHUNK_PAGES * PAGE_BYTES;
                if (used[half]) RJ_HIP(hipEventSynchronize(ev[half]));
                const void* const* pages = hc.pages + p0;
                parallel_for(np, 256, [&](size_t b, size_t e) {
                    for (size_t p = b; p < e; ++p) memcpy(s + p * PAGE_BYTES, pages[p], PAGE_BYTES);
                });
                RJ_HIP(hipMemcpyAsync(c.owned->as<uint8_t>() + p0 * PAGE_BYTES, s, np * PAGE_BYTES,
                                      hipMemcpyHostToDevice, ctx->stream));
                RJ_HIP(hipEventRecord(ev[half], ctx->stream));
                used[half] = true;
                half ^= 1;
            }
        }
        ctx->sync();
        for (TableColumn& c : t->cols)
            if (c.type != RJ_VARCHAR && !c.skipped) analyse_column(ctx, c, t->num_rows);
    } catch (...) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipEventDestroy(ev[0]);
        (void)hipEventDestroy(ev[1]);
        throw;
    }
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
    return t.release();
}

// Result pages -> caller-owned 8 KiB blocks (e.g. `new Page`, so the harness's
// Column::~Column, reference include/plan.h:95-99, can delete them).
void result_copy_pages(Result* r, uint64_t col, void* const* dst, uint64_t n_dst) {
    if (col >= r->cols.size()) throw_fmt(RJ_ERR_ARG, "column out of range");
    ResultColumn& c = r->cols[col];
    if (n_dst < c.n_pages) throw_fmt(RJ_ERR_ARG, "destination has too few pages");
    if (c.n_pages == 0) return;
    if (c.type == RJ_VARCHAR || !c.dev_pages) {
        const uint8_t* src = c.host_pages.data();
        parallel_for(c.n_pages, 256, [&](size_t b, size_t e) {
            for (size_t p = b; p < e; ++p) memcpy(dst[p], src + p * PAGE_BYTES, PAGE_BYTES);
        });
        return;
    }
    Context* ctx = r->ctx;
    uint8_t* stage = static_cast<uint8_t*>(ctx->staging(2 * CHUNK_PAGES * PAGE_BYTES));
    // D2H of chunk k+1 overlaps the host scatter of chunk k
    uint64_t   n_chunks = (c.n_pages + CHUNK_PAGES - 1) / CHUNK_PAGES;
    hipEvent_t ev[2];
    RJ_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    RJ_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    auto issue = [&](uint64_t k) {
        uint64_t p0 = k * CHUNK_PAGES, np = std::min<uint64_t>(CHUNK_PAGES, c.n_pages - p0);
        int      h = (int)(k & 1);
        RJ_HIP(hipMemcpyAsync(stage + (size_t)h * CHUNK_PAGES * PAGE_BYTES,
                              c.dev_pages->as<uint8_t>() + p0 * PAGE_BYTES, np * PAGE_BYTES,
                              hipMemcpyDeviceToHost, ctx->stream));
        RJ_HIP(hipEventRecord(ev[h], ctx->stream));
    };
    try {
        issue(0);
        for (uint64_t k = 0; k < n_chunks; ++k) {
            int h = (int)(k & 1);
            RJ_HIP(hipEventSynchronize(ev[h]));
            if (k + 1 < n_chunks) issue(k + 1);
            uint64_t       p0 = k * CHUNK_PAGES, np = std::min<uint64_t>(CHUNK_PAGES, c.n_pages - p0);
            const uint8_t* s = stage + (size_t)h * CHUNK_PAGES * PAGE_BYTES;
            void* const*   d = dst + p0;
            parallel_for(np, 256, [&](size_t b, size_t e) {
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

}  // namespace rj
