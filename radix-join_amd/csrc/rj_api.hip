// rj_api.hip — the extern "C" boundary (include/rj.h).  Every entry point turns
// C++ exceptions into status codes + rj_last_error().
#include <chrono>
#include <cstdlib>
#include <mutex>

#include "rj_internal.hpp"
#include "rj_xplan.hpp"

using namespace rj;

static std::mutex  g_err_mu;
static std::string g_create_error;

template <class F>
static int guarded(rj_context* ctx, F&& f) {
    try {
        f();
        return RJ_OK;
    } catch (const rj::Error& e) {
        if (ctx) ctx->last_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        if (ctx) ctx->last_error = "host allocation failed";
        return RJ_ERR_NOMEM;
    } catch (const std::exception& e) {
        if (ctx) ctx->last_error = e.what();
        return RJ_ERR_DEVICE;
    }
}

extern "C" {

int rj_abi_version(void) { return 3; }

int rj_context_create(rj_context** out, const rj_config* cfg) {
    if (!out) return RJ_ERR_ARG;
    *out = nullptr;
    int         code = RJ_OK;
    std::string msg;
    try {
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) {
            (void)hipGetLastError();
            throw rj::Error(RJ_ERR_NO_GPU,
                            "no HIP device: librj has no CPU fallback (the GPU path is the product)");
        }
        // one lane (single-device context) per local device; lane 0 is the handle
        std::vector<int> devs;
        if (cfg && cfg->n_devices > 0) {
            if (!cfg->devices) throw rj::Error(RJ_ERR_ARG, "n_devices > 0 but devices is NULL");
            for (int i = 0; i < cfg->n_devices; ++i) devs.push_back(cfg->devices[i]);
        } else {
            int dev = cfg ? cfg->device : -1;
            if (dev < 0) RJ_HIP(hipGetDevice(&dev));
            devs.push_back(dev);
        }
        if (devs.size() > 1 && cfg->stream)
            throw rj::Error(RJ_ERR_ARG, "a caller-provided stream needs a single-device context");
        auto make_lane = [&](int dev, bool first) {
            if (dev < 0 || dev >= n_dev) throw rj::Error(RJ_ERR_ARG, "device ordinal out of range");
            std::unique_ptr<rj_context> c(new rj_context());
            RJ_HIP(hipSetDevice(dev));
            c->device = dev;
            if (first && cfg && cfg->stream) {
                c->stream = static_cast<hipStream_t>(cfg->stream);
            } else {
                RJ_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
                c->own_stream = true;
            }
            c->prof.on = cfg && cfg->profile;
            c->prof.level = cfg && cfg->profile >= 2 ? 2 : 1;
            c->prof.stream = c->stream;
            c->radix_bits_override = cfg ? cfg->radix_bits : 0;
            c->tune.from_env();
            if (!c->radix_bits_override && c->tune.radix_bits > 0) c->radix_bits_override = c->tune.radix_bits;
            return c;
        };
        std::unique_ptr<rj_context> c = make_lane(devs[0], true);
        for (size_t i = 1; i < devs.size(); ++i) {
            rj_context* p = make_lane(devs[i], false).release();
            p->group = c.get();
            c->peers.push_back(p);
        }
        const int world = cfg && cfg->world_size > 0 ? cfg->world_size : (int)devs.size();
        if (world > 1 || (cfg && cfg->exchange != RJ_EXCHANGE_AUTO)) {
            std::vector<Context*> lanes;
            for (int i = 0; i < c->n_lanes(); ++i) lanes.push_back(c->lane(i));
            c->comm.reset(new Comm(lanes, world, cfg ? cfg->rank_base : 0, cfg ? cfg->exchange : 0,
                                   cfg ? cfg->comm_id : nullptr, c->tune.exchange_timeout_ms,
                                   c->tune.bringup_timeout_ms));
        }
        if (cfg && (cfg->flags & RJ_CTX_PREWARM))
            for (int i = 0; i < c->n_lanes(); ++i) c->lane(i)->prewarm();
        RJ_HIP(hipSetDevice(devs[0]));
        *out = c.release();
    } catch (const rj::Error& e) {
        code = e.code;
        msg = e.what();
    } catch (const std::exception& e) {
        code = RJ_ERR_DEVICE;
        msg = e.what();
    }
    if (code != RJ_OK) {
        std::lock_guard<std::mutex> g(g_err_mu);
        g_create_error = msg;
    }
    return code;
}

void rj_context_destroy(rj_context* ctx) {
    if (!ctx) return;
    // a per-device handle (rj_context_device) belongs to its group context: destroying it here
    // would free it a second time when the group goes (and lane 0 IS the group)
    if (ctx->group) {
        ctx->last_error = "rj_context_destroy: this is a per-device handle owned by its group context (ignored)";
        return;
    }
    delete ctx;
}

uint32_t rj_context_n_devices(const rj_context* ctx) { return ctx ? (uint32_t)ctx->n_lanes() : 0; }

rj_context* rj_context_device(rj_context* ctx, uint32_t i) {
    if (!ctx || i >= (uint32_t)ctx->n_lanes()) return nullptr;
    return static_cast<rj_context*>(ctx->lane((int)i));
}

int rj_comm_id_create(rj_comm_id* out) {
    if (!out) return RJ_ERR_ARG;
    try {
        Comm::make_id(out);
        return RJ_OK;
    } catch (const std::exception& e) {
        std::lock_guard<std::mutex> g(g_err_mu);
        g_create_error = e.what();  // rj_last_error(NULL)
        return RJ_ERR_DEVICE;
    }
}

const char* rj_last_error(const rj_context* ctx) {
    if (ctx) return ctx->last_error.c_str();
    std::lock_guard<std::mutex> g(g_err_mu);
    return g_create_error.c_str();
}

int rj_table_upload(rj_context* ctx, const rj_input* host, rj_table** out) {
    if (!ctx || !out) return RJ_ERR_ARG;
    *out = nullptr;
    return guarded(ctx, [&] {
        RJ_HIP(hipSetDevice(ctx->device));
        *out = static_cast<rj_table*>(table_upload(ctx, host));
    });
}

int rj_table_adopt_device(rj_context* ctx, uint64_t num_rows, uint64_t n_cols,
                          const int32_t* col_type, const void* const* dev_pages,
                          const uint64_t* n_pages, rj_table** out) {
    if (!ctx || !out || (n_cols && (!col_type || !dev_pages || !n_pages))) return RJ_ERR_ARG;
    *out = nullptr;
    return guarded(ctx, [&] {
        RJ_HIP(hipSetDevice(ctx->device));
        *out = static_cast<rj_table*>(
            table_adopt(ctx, num_rows, n_cols, col_type, dev_pages, n_pages));
    });
}

int rj_table_from_csv(rj_context* ctx, const char* text, uint64_t n_bytes, uint64_t n_cols, const int32_t* col_type,
                      const rj_filter_op* filter, uint64_t n_filter_ops, rj_table** out) {
    if (!ctx || !out || (n_filter_ops && !filter)) return RJ_ERR_ARG;
    *out = nullptr;
    return guarded(ctx, [&] {
        RJ_HIP(hipSetDevice(ctx->device));
        *out = static_cast<rj_table*>(table_from_csv(ctx, text, n_bytes, n_cols, col_type, filter, n_filter_ops));
    });
}

int rj_debug_parse_fp64(const char* field, uint64_t n, uint64_t* bits) {
    if (!bits || (n && !field)) return RJ_ERR_ARG;
    return rj::parse_fp64_host(field, n, bits);
}

uint64_t rj_table_num_rows(const rj_table* t) { return t ? t->num_rows : 0; }
uint64_t rj_table_col_pages(const rj_table* t, uint64_t col) { return table_col_pages(t, col); }

int rj_table_copy_pages(rj_context* ctx, const rj_table* t, uint64_t col, void* const* dst, uint64_t n_dst) {
    if (!ctx || !t || (n_dst && !dst)) return RJ_ERR_ARG;
    return guarded(ctx, [&] {
        RJ_HIP(hipSetDevice(ctx->device));
        table_copy_pages(ctx, t, col, dst, n_dst);
    });
}

void rj_table_release(rj_context* ctx, rj_table* t) {
    (void)ctx;
    delete t;
}

int rj_execute_resident(rj_context* ctx, const rj_plan* plan, rj_table* const* tables,
                        uint64_t n_tables, int32_t flags, rj_result** out) {
    if (!ctx || !plan || !out) return RJ_ERR_ARG;
    *out = nullptr;
    return guarded(ctx, [&] {
        RJ_HIP(hipSetDevice(ctx->device));
        std::vector<Table*> ts(n_tables);
        for (uint64_t i = 0; i < n_tables; ++i) {
            if (!tables[i]) throw rj::Error(RJ_ERR_ARG, "null table");
            ts[i] = tables[i];
        }
        *out = static_cast<rj_result*>(execute_plan(ctx, plan, ts.data(), n_tables, flags));
    });
}

int rj_execute(rj_context* ctx, const rj_plan* plan, rj_result** out) {
    if (!ctx || !plan || !out) return RJ_ERR_ARG;
    *out = nullptr;
    return guarded(ctx, [&] {
        RJ_HIP(hipSetDevice(ctx->device));
        if (plan->n_inputs && !plan->inputs) throw rj::Error(RJ_ERR_ARG, "plan has no inputs");
        // upload only the inputs some ScanNode reads
        std::vector<bool> used(plan->n_inputs, false);
        for (uint64_t i = 0; i < plan->n_nodes; ++i)
            if (plan->nodes[i].kind == RJ_NODE_SCAN) {
                if (plan->nodes[i].base_table_id >= plan->n_inputs)
                    throw rj::Error(RJ_ERR_ARG, "scan: bad base_table_id");
                used[plan->nodes[i].base_table_id] = true;
            }
        // ... and of those only the columns a ScanNode outputs (the harness hands over every
        // column of a scanned base table, reference tests/read_sql.cpp:1100-1107; the reference
        // decodes them all, src/build_table.cpp:317-434)
        std::vector<std::vector<bool>> col_used(plan->n_inputs);
        for (uint64_t i = 0; i < plan->n_inputs; ++i)
            col_used[i].assign(plan->inputs[i].n_cols, false);
        for (uint64_t i = 0; i < plan->n_nodes; ++i) {
            const rj_node& nd = plan->nodes[i];
            if (nd.kind != RJ_NODE_SCAN) continue;
            for (uint64_t k = 0; k < nd.n_out; ++k) {
                if (nd.out_idx[k] >= col_used[nd.base_table_id].size())
                    throw rj::Error(RJ_ERR_ARG, "scan: output attr out of range");
                col_used[nd.base_table_id][nd.out_idx[k]] = true;
            }
        }
        // a context that owns several devices shards the plan across them when it can
        if (ctx->n_lanes() > 1 && !ctx->group) {
            if (Result* r = execute_host_sharded(ctx, plan, used, col_used)) {
                *out = static_cast<rj_result*>(r);
                return;
            }
        }
        const bool  diag = ctx->tune.diag >= 2;
        auto        t0 = std::chrono::steady_clock::now();
        const uint64_t m0 = ctx->pool.n_malloc, tr0 = ctx->pool.n_trim;
        const double   mm0 = ctx->pool.malloc_ms;
        auto        ms_since = [](std::chrono::steady_clock::time_point t) {
            return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
        };
        if (ctx->tune.sync_upload > 0) {  // diagnostic: upload everything, then run the plan
            std::vector<std::unique_ptr<Table>> owned(plan->n_inputs);
            std::vector<Table*>                 ts(plan->n_inputs, nullptr);
            rj_input                            none{};
            for (uint64_t i = 0; i < plan->n_inputs; ++i) {
                owned[i].reset(table_upload(ctx, used[i] ? &plan->inputs[i] : &none,
                                            used[i] ? &col_used[i] : nullptr, /*borrow_varchar=*/true));
                ts[i] = owned[i].get();
            }
            double up = ms_since(t0);
            *out = static_cast<rj_result*>(execute_plan(ctx, plan, ts.data(), plan->n_inputs, 0));
            if (diag)
                fprintf(stderr, "[rj host] upload %.2f ms, then plan (device + host VARCHAR) %.2f ms\n", up,
                        ms_since(t0) - up);
            return;
        }
        // the plan starts at once; every ScanNode waits only for its own base table
        AsyncUpload  upl(ctx, plan, used, col_used);
        try {
            *out = static_cast<rj_result*>(execute_plan(ctx, plan, nullptr, plan->n_inputs, 0, &upl));
        } catch (...) {
            // kernels queued on the stream may still read the tables `upl` is about to release
            (void)hipStreamSynchronize(ctx->stream);
            throw;
        }
        if (diag)
            fprintf(stderr,
                    "[rj host] execute %.2f ms (plan walk blocked on uploads for %.2f ms); HBM cache: %llu "
                    "hipMalloc %.2f ms, %llu trims, %.1f GB in use, %.1f GB cached\n",
                    ms_since(t0), upl.wait_ms(), (unsigned long long)(ctx->pool.n_malloc - m0),
                    ctx->pool.malloc_ms - mm0, (unsigned long long)(ctx->pool.n_trim - tr0),
                    ctx->pool.bytes_in_use() / 1e9, ctx->pool.bytes_cached() / 1e9);
    });
}

uint64_t rj_result_num_rows(const rj_result* r) { return r ? r->num_rows : 0; }
uint64_t rj_result_num_cols(const rj_result* r) { return r ? r->cols.size() : 0; }
int32_t  rj_result_col_type(const rj_result* r, uint64_t c) {
    return (r && c < r->cols.size()) ? r->cols[c].type : -1;
}
uint64_t rj_result_col_pages(const rj_result* r, uint64_t c) {
    return (r && c < r->cols.size()) ? r->cols[c].n_pages : 0;
}

int rj_result_copy_pages(rj_result* r, uint64_t col, void* const* dst, uint64_t n_dst) {
    if (!r) return RJ_ERR_ARG;
    return guarded(static_cast<rj_context*>(r->ctx), [&] {
        RJ_HIP(hipSetDevice(r->ctx->device));
        result_copy_pages(r, col, dst, n_dst);
    });
}

const void* rj_result_device_pages(const rj_result* r, uint64_t c) {
    if (!r || c >= r->cols.size() || !r->cols[c].dev_pages) return nullptr;
    // a column gathered from several devices is not one contiguous run of rj_result_col_pages()
    // pages: fetch it with rj_result_copy_pages
    if (!r->cols[c].more.empty()) return nullptr;
    return r->cols[c].dev_pages->p;
}

void rj_result_free(rj_result* r) { delete r; }

int rj_execute_sharded(rj_context* ctx, const rj_plan* plan, rj_table* const* tables,
                       uint64_t n_inputs, int32_t flags, rj_result** out) {
    if (!ctx || !plan || !out || (n_inputs && !tables)) return RJ_ERR_ARG;
    const int nl = ctx->n_lanes();
    for (int l = 0; l < nl; ++l) out[l] = nullptr;
    return guarded(ctx, [&] {
        if (ctx->group) throw rj::Error(RJ_ERR_ARG, "rj_execute_sharded wants the group context, not one of its devices");
        std::vector<Table*>  ts((size_t)nl * n_inputs);
        for (size_t i = 0; i < ts.size(); ++i) ts[i] = tables[i];
        std::vector<Result*> rs((size_t)nl, nullptr);
        execute_sharded(ctx, plan, ts.data(), n_inputs, flags, rs.data());
        for (int l = 0; l < nl; ++l) out[l] = static_cast<rj_result*>(rs[l]);
    });
}

int rj_plan_shardable(const rj_plan* plan, char* why, size_t why_cap) {
    std::string w;
    bool        ok = false;
    try {
        ok = plan_shardable(plan, &w);
    } catch (const std::exception& e) {
        w = e.what();
    }
    if (why && why_cap) {
        strncpy(why, ok ? "" : (w.empty() ? "malformed plan" : w.c_str()), why_cap - 1);
        why[why_cap - 1] = 0;
    }
    return ok ? 1 : 0;
}

int rj_exchange_plan(uint32_t world, uint32_t subs, uint32_t rank, const uint64_t* counts, uint64_t* send_off,
                     uint64_t* send_cnt, uint64_t* recv_off, uint64_t* recv_cnt, uint32_t* seg_begin,
                     uint32_t* seg_end, uint32_t* part_off, uint64_t* n_recv) {
    if (!counts) return RJ_ERR_ARG;
    try {
        ExchangePlan p;
        exchange_plan(world, subs, rank, counts, p);
        auto put = [](auto* dst, const auto& v) {
            if (dst) std::copy(v.begin(), v.end(), dst);
        };
        put(send_off, p.send_off);
        put(send_cnt, p.send_cnt);
        put(recv_off, p.recv_off);
        put(recv_cnt, p.recv_cnt);
        put(seg_begin, p.seg_begin);
        put(seg_end, p.seg_end);
        put(part_off, p.part_off);
        if (n_recv) *n_recv = p.n_recv;
        return RJ_OK;
    } catch (const rj::Error& e) {
        std::lock_guard<std::mutex> g(g_err_mu);
        g_create_error = e.what();  // rj_last_error(NULL)
        return e.code;
    } catch (const std::exception& e) {
        std::lock_guard<std::mutex> g(g_err_mu);
        g_create_error = e.what();
        return RJ_ERR_DEVICE;
    }
}

int rj_shard_partition(rj_context* ctx, const rj_table* t, uint64_t key_col, uint64_t carry_col,
                       uint32_t n_ranks, rj_tuples* out, uint64_t* counts) {
    if (!ctx || !t || !out || !counts) return RJ_ERR_ARG;
    return guarded(ctx, [&] {
        RJ_HIP(hipSetDevice(ctx->device));
        shard_partition(ctx, t, key_col, carry_col, n_ranks, out, counts);
    });
}

int rj_join_tuples(rj_context* ctx, const rj_tuples* build, const rj_tuples* probe,
                   uint32_t skip_rank_bits, int32_t flags, rj_result** out) {
    if (!ctx || !build || !probe || !out) return RJ_ERR_ARG;
    *out = nullptr;
    return guarded(ctx, [&] {
        RJ_HIP(hipSetDevice(ctx->device));
        *out = static_cast<rj_result*>(join_tuples(ctx, build, probe, skip_rank_bits, flags));
    });
}

int rj_profile_read(rj_context* ctx, rj_kernel_stat* out, uint64_t cap, uint64_t* n) {
    if (!ctx || !n) return RJ_ERR_ARG;
    return guarded(ctx, [&] {
        ctx->prof.drain();
        uint64_t k = 0;
        for (const std::string& name : ctx->prof.order) {
            if (k < cap && out) {
                memset(&out[k], 0, sizeof out[k]);
                strncpy(out[k].name, name.c_str(), sizeof(out[k].name) - 1);
                out[k].launches = ctx->prof.totals[name].launches;
                out[k].total_ms = ctx->prof.totals[name].ms;
            }
            ++k;
        }
        *n = k;
    });
}

void rj_profile_reset(rj_context* ctx) {
    if (!ctx) return;
    (void)guarded(ctx, [&] { ctx->prof.reset(); });
}

int rj_device_query(rj_context* ctx, rj_device_info* out) {
    if (!ctx || !out) return RJ_ERR_ARG;
    return guarded(ctx, [&] {
        hipDeviceProp_t p;
        RJ_HIP(hipGetDeviceProperties(&p, ctx->device));
        memset(out, 0, sizeof *out);
        strncpy(out->name, p.name, sizeof(out->name) - 1);
        strncpy(out->arch, p.gcnArchName, sizeof(out->arch) - 1);
        out->compute_units = p.multiProcessorCount;
        out->wavefront = p.warpSize;
        out->hbm_bytes = p.totalGlobalMem;
        out->lds_per_cu = p.maxSharedMemoryPerMultiProcessor;
        int n = 0;
        RJ_HIP(hipGetDeviceCount(&n));
        out->device_count = n;
    });
}

}  // extern "C"
