// rj_context.hip — context, HBM block cache, HIP-event profiler.
#include <chrono>
#include <cstdlib>

#include "rj_internal.hpp"

namespace rj {

[[noreturn]] void launch_failed(const char* kernel, const char* what, bool unsupported) {
    throw_fmt(unsupported ? RJ_ERR_UNSUPPORTED : RJ_ERR_DEVICE, "launch %s: %s", kernel, what);
}

static int env_int(const char* name, int def) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : def;
}

void Tuning::from_env() {
    blocked_mid = env_int("RJ_TUNE_BLOCKED_MID", blocked_mid);
    mall_chunk = env_int("RJ_TUNE_MALL_CHUNK", mall_chunk);
    radix_bits = env_int("RJ_TUNE_RADIX_BITS", radix_bits);
    p1_bits = env_int("RJ_TUNE_P1_BITS", p1_bits);
    fine = env_int("RJ_TUNE_FINE", fine);
    pack = env_int("RJ_TUNE_PACK", pack);
    aos3 = env_int("RJ_TUNE_AOS3", aos3);
    aos_mid = env_int("RJ_TUNE_AOS_MID", aos_mid);
    tpg1 = env_int("RJ_TUNE_TPG1", tpg1);
    xcd_split = env_int("RJ_TUNE_XCD_SPLIT", xcd_split);
    tpg2 = env_int("RJ_TUNE_TPG2", tpg2);
    packed_side = env_int("RJ_TUNE_PACKED_SIDE", packed_side);
    xcd_min_rows = env_int("RJ_TUNE_XCD_MIN_ROWS", xcd_min_rows);
    bcast = env_int("RJ_TUNE_BCAST", bcast);
    diag = env_int("RJ_DIAG", diag);
    varchar_dev_rows = env_int("RJ_TUNE_VARCHAR_DEV", varchar_dev_rows);
    vkey_hash_bits = env_int("RJ_DEBUG_VKEY_HASH_BITS", vkey_hash_bits);
    sync_upload = env_int("RJ_SYNC_UPLOAD", sync_upload);
    wide_carry = env_int("RJ_TUNE_WIDE_CARRY", wide_carry);
    fold_owner = env_int("RJ_TUNE_FOLD_OWNER", fold_owner);
    exchange_timeout_ms = env_int("RJ_EXCHANGE_TIMEOUT_MS", exchange_timeout_ms);
    bringup_timeout_ms = env_int("RJ_BRINGUP_TIMEOUT_MS", bringup_timeout_ms);
    debug_shard_fail = env_int("RJ_DEBUG_SHARD_FAIL", debug_shard_fail);
    debug_shard_fail_rank = env_int("RJ_DEBUG_SHARD_FAIL_RANK", debug_shard_fail_rank);
}

// ---------------------------------------------------------------- DevPool --
static size_t round_size(size_t n) {
    const size_t small = 256, big = size_t(2) << 20;
    if (n < (size_t(1) << 20)) return (n + small - 1) / small * small;
    return (n + big - 1) / big * big;
}

void* DevPool::alloc(size_t bytes) {
    size_t need = round_size(bytes ? bytes : 1);
    int    best = -1;
    for (size_t i = 0; i < blocks_.size(); ++i) {
        Block& b = blocks_[i];
        if (!b.free || b.size < need) continue;
        if (b.size > need + need / 2 + (size_t(4) << 20)) continue;  // don't burn huge blocks
        if (best < 0 || b.size < blocks_[best].size) best = (int)i;
    }
    if (best >= 0) {
        blocks_[best].free = false;
        in_use_ += blocks_[best].size;
        cached_ -= blocks_[best].size;
        return blocks_[best].p;
    }
    void*      p = nullptr;
    auto       t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&p, need);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        trim();
        ++n_trim;
        e = hipMalloc(&p, need);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            throw_fmt(RJ_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", need, hipGetErrorString(e));
        }
    }
    ++n_malloc;
    malloc_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    blocks_.push_back({p, need, false});
    in_use_ += need;
    return p;
}

void DevPool::release(void* p) {
    if (!p) return;
    for (Block& b : blocks_)
        if (b.p == p && !b.free) {
            b.free = true;
            in_use_ -= b.size;
            cached_ += b.size;
            return;
        }
}

void DevPool::trim() {
    std::vector<Block> keep;
    for (Block& b : blocks_) {
        if (b.free) {
            (void)hipFree(b.p);
            cached_ -= b.size;
        } else {
            keep.push_back(b);
        }
    }
    blocks_.swap(keep);
}

DevPool::~DevPool() {
    for (Block& b : blocks_) (void)hipFree(b.p);
}

Buf::Buf(Context* c, size_t n) : ctx(c), bytes(n) { p = c->pool.alloc(n); }
Buf::~Buf() {
    if (ctx && p) ctx->pool.release(p);
}

// --------------------------------------------------------------- Profiler --
hipEvent_t Profiler::get_event() {
    if (!spare_.empty()) {
        hipEvent_t e = spare_.back();
        spare_.pop_back();
        return e;
    }
    hipEvent_t e;
    RJ_HIP(hipEventCreate(&e));
    return e;
}

// Level 1 times only the kernels that move the data, level 2 every launch.  The events are
// recorded by the launch itself (start/stop timestamps of the dispatch), not by separate
// event packets around it.
static bool hot_kernel(const char* n) {
    return !strncmp(n, "pass", 4) || !strncmp(n, "join_build", 10) || !strncmp(n, "exchange", 8);
}

bool Profiler::timed(const char* name, hipEvent_t* start, hipEvent_t* stop) {
    if (!on || (level < 2 && !hot_kernel(name))) return false;
    Rec r{name, get_event(), get_event()};
    open_.push_back(r);
    *start = r.a;
    *stop = r.b;
    return true;
}

void Profiler::drain() {
    if (open_.empty()) return;
    RJ_HIP(hipStreamSynchronize(stream));
    for (Rec& r : open_) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto it = totals.find(r.name);
            if (it == totals.end()) {
                order.push_back(r.name);
                it = totals.emplace(r.name, Tot{}).first;
            }
            it->second.launches++;
            it->second.ms += ms;
        } else {
            (void)hipGetLastError();
        }
        spare_.push_back(r.a);
        spare_.push_back(r.b);
    }
    open_.clear();
}

void Profiler::reset() {
    drain();
    totals.clear();
    order.clear();
}

Profiler::~Profiler() {
    for (Rec& r : open_) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    for (hipEvent_t e : spare_) (void)hipEventDestroy(e);
}

// ---------------------------------------------------------------- Context --
void* Context::staging(size_t bytes) {
    if (bytes > pinned_bytes) {
        if (pinned) (void)hipHostFree(pinned);
        pinned = nullptr;
        pinned_bytes = 0;
        RJ_HIP(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
        pinned_bytes = bytes;
    }
    return pinned;
}

void* Context::small_pinned() {
    if (!pinned_small) RJ_HIP(hipHostMalloc(&pinned_small, SMALL_PINNED, hipHostMallocDefault));
    return pinned_small;
}

int Context::compute_units() {
    if (n_cu <= 0) {
        int v = 0;
        RJ_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device));
        n_cu = v > 0 ? v : 256;
    }
    return n_cu;
}

// What the first rj_execute of a fresh context would otherwise pay (job/1a: 44 ms against 1.4 ms
// for the calls after it): pinned staging in both directions, the upload stream, the host worker
// threads, and the code objects of the kernels, which HIP loads when a kernel of a translation
// unit is first launched.
void Context::prewarm() {
    RJ_HIP(hipSetDevice(device));
    constexpr size_t STAGE = (size_t)2 * 4096 * PAGE_BYTES;  // rj_table.hip: two 32 MiB halves
    (void)staging(STAGE);
    (void)upload_staging(STAGE);
    (void)small_pinned();
    (void)upload_stream();
    (void)compute_units();
    parallel_for(4096, 1, [](size_t, size_t) {});  // spins up the host pool
    BufP         b = buf(256);
    const Launch L = launch();
    // first use of the copy engines in both directions, on both streams
    memset(pinned_up, 0, 256);
    RJ_HIP(hipMemcpyAsync(b->p, pinned_up, 256, hipMemcpyHostToDevice, copy_stream));
    RJ_HIP(hipStreamSynchronize(copy_stream));
    RJ_HIP(hipMemcpyAsync(pinned, b->p, 256, hipMemcpyDeviceToHost, stream));
    RJ_HIP(hipMemcpyAsync(pinned_small, b->p, 64, hipMemcpyDeviceToHost, stream));
    RJ_HIP(hipMemsetAsync(b->p, 0, 256, stream));
    launch_scan_bins(L, b->as<uint32_t>(), 1, b->as<uint32_t>() + 8, nullptr);  // rj_kernels.hip's code object
    prewarm_varchar_dev(L, b->as<uint32_t>());                                    // rj_varchar_dev.hip's
    sync();
}

hipStream_t Context::aux_stream() {
    if (!aux) RJ_HIP(hipStreamCreateWithFlags(&aux, hipStreamNonBlocking));
    return aux;
}

hipStream_t Context::upload_stream() {
    if (!copy_stream) RJ_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    return copy_stream;
}

void* Context::upload_staging(size_t bytes) {
    if (bytes > pinned_up_bytes) {
        if (pinned_up) (void)hipHostFree(pinned_up);
        pinned_up = nullptr;
        pinned_up_bytes = 0;
        RJ_HIP(hipHostMalloc(&pinned_up, bytes, hipHostMallocDefault));
        pinned_up_bytes = bytes;
    }
    return pinned_up;
}

Context::~Context() {
    // the exchange transport references the lanes: it goes first, then the lanes themselves
    comm.reset();
    for (Context* p : peers) delete static_cast<rj_context*>(p);
    peers.clear();
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    if (aux) {
        (void)hipStreamSynchronize(aux);
        (void)hipStreamDestroy(aux);
    }
    if (copy_stream) {
        (void)hipStreamSynchronize(copy_stream);
        (void)hipStreamDestroy(copy_stream);
    }
    if (pinned_up) (void)hipHostFree(pinned_up);
    if (pinned) (void)hipHostFree(pinned);
    if (pinned_small) (void)hipHostFree(pinned_small);
    if (own_stream && stream) (void)hipStreamDestroy(stream);
}

}  // namespace rj
