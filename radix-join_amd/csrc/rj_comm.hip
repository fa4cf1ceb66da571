// rj_comm.hip — exchange transports of the sharded join (see rj_comm.hpp).
#include "rj_comm.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <set>

#include "rj_internal.hpp"

// Types and prototypes only: the library itself is dlopen'ed (no link-time dependency).  A
// single-GPU build box without the RCCL headers still compiles: the handful of declarations this
// file needs is restated below (values as in nccl.h; checked against the header where it exists).
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4,
               ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId*);
ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclCommAbort(ncclComm_t);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
ncclResult_t ncclSend(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclRecv(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclAllGather(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
const char*  ncclGetErrorString(ncclResult_t);
}
#endif
static_assert(ncclUint8 == 1 && ncclUint64 == 5 && ncclSuccess == 0, "nccl enum values");

namespace rj {

namespace {

struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId)    GetUniqueId = nullptr;
    decltype(&ncclCommInitRank)   CommInitRank = nullptr;
    decltype(&ncclCommDestroy)    CommDestroy = nullptr;
    decltype(&ncclCommAbort)      CommAbort = nullptr;  // optional
    decltype(&ncclGroupStart)     GroupStart = nullptr;
    decltype(&ncclGroupEnd)       GroupEnd = nullptr;
    decltype(&ncclSend)           Send = nullptr;
    decltype(&ncclRecv)           Recv = nullptr;
    decltype(&ncclAllGather)      AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

bool comm_trace() {
    static const bool on = [] {
        const char* v = getenv("RJ_DIAG");
        return v && atoi(v) >= 2;
    }();
    return on;
}
#define RJ_COMM_TRACE(...)                        \
    do {                                          \
        if (comm_trace()) {                       \
            fprintf(stderr, "[rj comm] " __VA_ARGS__); \
            fputc('\n', stderr);                  \
            fflush(stderr);                       \
        }                                         \
    } while (0)

Rccl& rccl() {
    static Rccl       r;
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    if (r.h) return r;
    // The soname FIRST: a process that already holds a librccl (PyTorch bundles one, built against
    // the HIP runtime it also bundles) must bind to THAT copy.  Opening another copy by path pulls
    // a second RCCL — and through its DT_NEEDED possibly a second HIP runtime — into the process;
    // a communicator brought up in the second copy cannot see the first runtime's devices and
    // streams ("the hang of round 2").  RJ_RCCL_PATH overrides the search.
    const char* names[] = {getenv("RJ_RCCL_PATH"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void*       h = nullptr;
    for (const char* n : names) {
        if (!n || !*n) continue;
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) {
            RJ_COMM_TRACE("dlopen(%s) ok", n);
            break;
        }
    }
    if (!h) throw_fmt(RJ_ERR_DEVICE, "cannot load librccl (%s): multi-process sharding needs RCCL", dlerror());
    auto sym = [&](const char* n, bool required = true) {
        void* p = dlsym(h, n);
        if (!p && required) throw_fmt(RJ_ERR_DEVICE, "librccl lacks %s", n);
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort", false));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (comm_trace()) {
        Dl_info di{};
        if (dladdr(reinterpret_cast<void*>(r.CommInitRank), &di) && di.dli_fname)
            RJ_COMM_TRACE("ncclCommInitRank comes from %s", di.dli_fname);
        if (dladdr(reinterpret_cast<void*>(&hipStreamCreate), &di) && di.dli_fname)
            RJ_COMM_TRACE("librj's HIP runtime is %s", di.dli_fname);
    }
    r.h = h;
    return r;
}

#define RJ_NCCL(expr)                                                                          \
    do {                                                                                       \
        ncclResult_t _r = (expr);                                                              \
        if (_r != ncclSuccess)                                                                 \
            ::rj::throw_fmt(RJ_ERR_DEVICE, "%s failed: %s", #expr, rccl().GetErrorString(_r)); \
    } while (0)

using Clock = std::chrono::steady_clock;

}  // namespace

// One helper thread per Comm runs the host calls that may block on a peer, one at a time.  A
// call that outlives its deadline leaves the thread stuck inside RCCL: the Comm gives it up (the
// thread keeps the state it touches alive through the shared_ptr) and never uses it again.
struct Comm::Worker {
    std::mutex              mu;
    std::condition_variable cv;
    std::function<void()>   job;
    bool                    has_job = false, done = false, quit = false;
    std::exception_ptr      err;
    std::thread             th;
    static void run(std::shared_ptr<Worker> w) {
        std::unique_lock<std::mutex> lk(w->mu);
        for (;;) {
            w->cv.wait(lk, [&] { return w->has_job || w->quit; });
            if (w->quit) return;
            std::function<void()> fn = std::move(w->job);
            w->has_job = false;
            lk.unlock();
            std::exception_ptr e;
            try {
                fn();
            } catch (...) {
                e = std::current_exception();
            }
            lk.lock();
            w->err = e;
            w->done = true;
            w->cv.notify_all();
        }
    }
};

void Comm::check_alive() const {
    if (failed_)
        throw_fmt(RJ_ERR_DEVICE, "the exchange transport failed earlier (%s): this context can only be destroyed",
                  failed_what_.c_str());
}

void Comm::mark_failed(const char* what, const char* why) {
    if (failed_) return;
    failed_ = true;
    failed_what_ = std::string(what) + " " + why;
    // unblock whatever still waits inside RCCL (kernels spinning for a peer, a host call
    // waiting for a connection): abort is the one call that may be made from another thread
    if (mode_ == RCCL) {
        Rccl& R = rccl();
        for (void*& c : nccl_)
            if (c && R.CommAbort) {
                RJ_COMM_TRACE("ncclCommAbort after: %s", failed_what_.c_str());
                (void)R.CommAbort(static_cast<ncclComm_t>(c));
                c = nullptr;
            }
    }
}

void Comm::fail(const char* what, const char* why, int ms) {
    mark_failed(what, why);
    throw_fmt(RJ_ERR_DEVICE,
              "sharded join, rank %d: %s %s within %d ms (%s) — a peer rank failed, hangs or "
              "never started; the job is lost, this process should exit",
              rank_base_, what, why, ms ? ms : timeout_ms_, ms ? "RJ_BRINGUP_TIMEOUT_MS" : "RJ_EXCHANGE_TIMEOUT_MS");
}

void Comm::bounded(const char* what, std::function<void()> fn, int ms) {
    check_alive();
    if (!worker_) {
        worker_ = std::make_shared<Worker>();
        worker_->th = std::thread(Worker::run, worker_);
    }
    Worker&                      w = *worker_;
    std::unique_lock<std::mutex> lk(w.mu);
    w.job = std::move(fn);
    w.has_job = true;
    w.done = false;
    w.err = nullptr;
    w.cv.notify_all();
    const auto deadline = Clock::now() + std::chrono::milliseconds(ms ? ms : timeout_ms_);
    if (!w.cv.wait_until(lk, deadline, [&] { return w.done; })) {
        lk.unlock();
        mark_failed(what, "did not return");  // aborts the communicators: the blocked call then errors out
        lk.lock();
        if (!w.cv.wait_for(lk, std::chrono::seconds(3), [&] { return w.done; })) {
            // still stuck inside RCCL (e.g. bring-up: no communicator to abort yet): the helper is
            // given up — detached, it keeps its own state alive through its shared_ptr
            lk.unlock();
            w.th.detach();
            worker_.reset();
        } else {
            lk.unlock();
        }
        fail(what, "did not return", ms);
    }
    std::exception_ptr e = w.err;
    lk.unlock();
    if (e) std::rethrow_exception(e);
}

void Comm::wait_event(int lane, hipEvent_t ev, const char* what) {
    check_alive();
    RJ_HIP(hipSetDevice(lanes_[lane]->device));
    const auto deadline = Clock::now() + std::chrono::milliseconds(timeout_ms_);
    for (uint32_t spins = 0;; ++spins) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) RJ_HIP(e);
        (void)hipGetLastError();
        if (Clock::now() > deadline) fail(what, "did not complete");
        // the first microseconds are polled hot (an exchange at 8 ranks takes a few ms), then yield
        if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(spins > 20000 ? 500 : 20));
    }
}

void Comm::wait_stream(int lane, const char* what) {
    check_alive();
    RJ_HIP(hipSetDevice(lanes_[lane]->device));
    const auto deadline = Clock::now() + std::chrono::milliseconds(timeout_ms_);
    for (uint32_t spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(xfer_[lane]);
        if (e == hipSuccess) return;
        if (e != hipErrorNotReady) RJ_HIP(e);
        (void)hipGetLastError();
        if (Clock::now() > deadline) fail(what, "did not complete");
        if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(spins > 20000 ? 500 : 20));
    }
}

void Comm::make_id(rj_comm_id* out) {
    static_assert(sizeof(ncclUniqueId) == RJ_COMM_ID_BYTES, "rj_comm_id is an ncclUniqueId");
    ncclUniqueId id;
    RJ_NCCL(rccl().GetUniqueId(&id));
    memcpy(out->bytes, id.internal, RJ_COMM_ID_BYTES);
}

Comm::Comm(std::vector<Context*> lanes, int world, int rank_base, int mode, const rj_comm_id* id, int timeout_ms,
           int bringup_ms)
    : lanes_(std::move(lanes)), world_(world), rank_base_(rank_base) {
    const int nl = (int)lanes_.size();
    if (timeout_ms > 0) timeout_ms_ = timeout_ms;
    if (nl < 1 || world < nl || rank_base < 0 || rank_base + nl > world)
        throw_fmt(RJ_ERR_ARG, "bad rank layout: %d local ranks from %d in a world of %d", nl, rank_base, world);
    if (world & (world - 1)) throw_fmt(RJ_ERR_UNSUPPORTED, "world size must be a power of two");
    const bool all_local = nl == world;
    if (mode == 0) mode = all_local ? P2P : RCCL;
    if (mode == P2P && !all_local)
        throw_fmt(RJ_ERR_ARG, "peer-to-peer exchange needs every rank in this process (world %d, local %d)", world, nl);
    if (mode == RCCL) {
        // two communicator ranks on one device inside one group fail or hang in RCCL: virtual
        // ranks (a repeated ordinal, the single-GPU test set-up) are for the peer-copy transport
        std::set<int> seen;
        for (Context* c : lanes_)
            if (!seen.insert(c->device).second)
                throw_fmt(RJ_ERR_ARG, "RCCL exchange: device %d appears twice in `devices` (virtual ranks need RJ_EXCHANGE_P2P)",
                          c->device);
    }
    mode_ = (Mode)mode;
    xfer_.assign(nl, nullptr);
    sent_.assign(nl, nullptr);
    cnt_dev_.assign(nl, nullptr);
    nccl_.assign(nl, nullptr);
    for (int l = 0; l < nl; ++l) {
        RJ_HIP(hipSetDevice(lanes_[l]->device));
        RJ_HIP(hipStreamCreateWithFlags(&xfer_[l], hipStreamNonBlocking));
        RJ_HIP(hipEventCreateWithFlags(&sent_[l], hipEventDisableTiming));
    }
    if (mode_ == P2P) {
        // direct xGMI copies between every pair of local devices
        for (int a = 0; a < nl; ++a)
            for (int b = 0; b < nl; ++b) {
                const int da = lanes_[a]->device, db = lanes_[b]->device;
                if (da == db) continue;
                int can = 0;
                RJ_HIP(hipDeviceCanAccessPeer(&can, da, db));
                if (!can) continue;
                RJ_HIP(hipSetDevice(da));
                hipError_t e = hipDeviceEnablePeerAccess(db, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) RJ_HIP(e);
                (void)hipGetLastError();
            }
    } else {
        if (!id) throw_fmt(RJ_ERR_ARG, "RCCL exchange needs an rj_comm_id shared by all ranks (rj_comm_id_create)");
        ncclUniqueId nid;
        memcpy(nid.internal, id->bytes, RJ_COMM_ID_BYTES);
        Rccl& R = rccl();
        // communicator bring-up rendezvous with every other rank of the job: bounded like every
        // other wait (a rank that never starts must not hold the others forever)
        RJ_COMM_TRACE("ncclCommInitRank: rank %d.. of %d (%d local)", rank_base_, world_, nl);
        std::vector<void*>* out = &nccl_;
        const int           wd = world_, rb = rank_base_;
        std::vector<int>    devs;
        for (Context* c : lanes_) devs.push_back(c->device);
        auto slots = std::make_shared<std::vector<void*>>((size_t)nl, nullptr);
        try {
            bounded("communicator bring-up (ncclCommInitRank)", [slots, nid, wd, rb, devs, nl, &R] {
                if (nl > 1) RJ_NCCL(R.GroupStart());
                for (int l = 0; l < nl; ++l) {
                    RJ_HIP(hipSetDevice(devs[l]));
                    ncclComm_t c = nullptr;
                    RJ_NCCL(R.CommInitRank(&c, wd, nid, rb + l));
                    (*slots)[l] = c;
                }
                if (nl > 1) RJ_NCCL(R.GroupEnd());
            }, bringup_ms);
        } catch (...) {
            for (int l = 0; l < nl; ++l) {
                (void)hipSetDevice(lanes_[l]->device);
                if (sent_[l]) (void)hipEventDestroy(sent_[l]);
                if (xfer_[l]) (void)hipStreamDestroy(xfer_[l]);
            }
            if (worker_) {
                {
                    std::lock_guard<std::mutex> g(worker_->mu);
                    worker_->quit = true;
                    worker_->cv.notify_all();
                }
                if (worker_->th.joinable()) worker_->th.join();
            }
            throw;
        }
        *out = *slots;
        RJ_COMM_TRACE("communicator up");
    }
    RJ_HIP(hipSetDevice(lanes_[0]->device));
}

Comm::~Comm() {
    for (size_t l = 0; l < lanes_.size(); ++l) {
        (void)hipSetDevice(lanes_[l]->device);
        // a failed transport was aborted: its streams may never drain, do not wait for them
        if (xfer_[l] && !failed_) (void)hipStreamSynchronize(xfer_[l]);
        if (nccl_[l]) (void)rccl().CommDestroy(static_cast<ncclComm_t>(nccl_[l]));
        if (cnt_dev_[l]) (void)hipFree(cnt_dev_[l]);
        if (sent_[l]) (void)hipEventDestroy(sent_[l]);
        if (xfer_[l] && !failed_) (void)hipStreamDestroy(xfer_[l]);
    }
    if (cnt_host_) (void)hipHostFree(cnt_host_);
    if (worker_) {
        {
            std::lock_guard<std::mutex> g(worker_->mu);
            worker_->quit = true;
            worker_->cv.notify_all();
        }
        if (worker_->th.joinable()) worker_->th.join();
    }
    if (!lanes_.empty()) (void)hipSetDevice(lanes_[0]->device);
}

void Comm::allgather_u64(const std::vector<std::vector<uint64_t>>& vals, size_t k,
                         std::vector<std::vector<uint64_t>>& all) {
    check_alive();
    const int nl = n_local();
    all.assign(world_, std::vector<uint64_t>(k, 0));
    if (mode_ == P2P) {
        for (int l = 0; l < nl; ++l) all[rank_base_ + l] = vals[l];
        return;
    }
    Rccl& R = rccl();
    // Sized once for the largest tensor a join gathers (2 sides x 512 stage-A partitions + a status
    // word): growing the buffers later would take a hipFree, which waits for EVERY stream of the
    // device — stage A's scatters, which the exchange is to overlap — and, behind a stalled stream,
    // would sit there without a bound.
    const size_t kcap = std::max<size_t>(k, 2 * 512 + 1);
    const size_t need = (size_t)(world_ + 1) * kcap * 8;
    if (need > cnt_cap_) {
        for (int l = 0; l < nl; ++l) {
            RJ_HIP(hipSetDevice(lanes_[l]->device));
            if (cnt_dev_[l]) RJ_HIP(hipFree(cnt_dev_[l]));
            cnt_dev_[l] = nullptr;
            RJ_HIP(hipMalloc(&cnt_dev_[l], need));
        }
        if (cnt_host_) RJ_HIP(hipHostFree(cnt_host_));
        cnt_host_ = nullptr;
        RJ_HIP(hipHostMalloc(&cnt_host_, (size_t)(world_ + nl) * kcap * 8, hipHostMallocDefault));
        cnt_cap_ = need;
    }
    // (pinned both ways and on the exchange streams: a synchronous copy, or one from pageable
    // memory, would make the host wait behind the compute streams)
    uint64_t* hin = static_cast<uint64_t*>(cnt_host_) + (size_t)world_ * k;
    for (int l = 0; l < nl; ++l) {
        RJ_HIP(hipSetDevice(lanes_[l]->device));
        memcpy(hin + (size_t)l * k, vals[l].data(), k * 8);
        RJ_HIP(hipMemcpyAsync(cnt_dev_[l], hin + (size_t)l * k, k * 8, hipMemcpyHostToDevice, xfer_[l]));
    }
    struct Op {  // (by value into the helper thread, as in all_to_all)
        int         dev;
        uint64_t*   base;
        ncclComm_t  comm;
        hipStream_t stream;
    };
    auto ops = std::make_shared<std::vector<Op>>();
    for (int l = 0; l < nl; ++l)
        ops->push_back(Op{lanes_[l]->device, static_cast<uint64_t*>(cnt_dev_[l]), static_cast<ncclComm_t>(nccl_[l]), xfer_[l]});
    Rccl* const rp = &R;
    bounded("count all-gather (ncclAllGather)", [ops, rp, k] {
        if (ops->size() > 1) RJ_NCCL(rp->GroupStart());
        for (const Op& op : *ops) {
            RJ_HIP(hipSetDevice(op.dev));
            RJ_NCCL(rp->AllGather(op.base, op.base + k, k, ncclUint64, op.comm, op.stream));
        }
        if (ops->size() > 1) RJ_NCCL(rp->GroupEnd());
    });
    RJ_HIP(hipSetDevice(lanes_[0]->device));
    RJ_HIP(hipMemcpyAsync(cnt_host_, static_cast<uint64_t*>(cnt_dev_[0]) + k, (size_t)world_ * k * 8,
                          hipMemcpyDeviceToHost, xfer_[0]));
    for (int l = 0; l < nl; ++l) wait_stream(l, "count all-gather");
    const uint64_t* host = static_cast<const uint64_t*>(cnt_host_);
    for (int r = 0; r < world_; ++r) all[r].assign(host + (size_t)r * k, host + (size_t)(r + 1) * k);
    RJ_HIP(hipSetDevice(lanes_[0]->device));
}

void Comm::all_to_all(const std::vector<std::vector<XferSpec>>& specs, const std::vector<hipEvent_t>& ready,
                      std::vector<hipEvent_t>& done, const char* name) {
    check_alive();
    const int nl = n_local();
    // profiling contexts: start/stop events on each lane's exchange stream
    std::vector<hipEvent_t> t0((size_t)nl, nullptr), t1((size_t)nl, nullptr);
    for (int l = 0; l < nl; ++l) {
        RJ_HIP(hipSetDevice(lanes_[l]->device));  // (events belong to the device they are made on)
        if (!lanes_[l]->prof.timed(name, &t0[l], &t1[l])) t0[l] = t1[l] = nullptr;
    }
    if (mode_ == P2P) {
        // a copy touches the sender's AND the receiver's buffer: every exchange stream waits for
        // every lane's `ready`
        for (int l = 0; l < nl; ++l) {
            RJ_HIP(hipSetDevice(lanes_[l]->device));
            for (int o = 0; o < nl; ++o) RJ_HIP(hipStreamWaitEvent(xfer_[l], ready[o], 0));
            if (t0[l]) RJ_HIP(hipEventRecord(t0[l], xfer_[l]));
        }
        for (int s = 0; s < nl; ++s) {
            RJ_HIP(hipSetDevice(lanes_[s]->device));
            for (size_t a = 0; a < specs[s].size(); ++a) {
                const XferSpec& S = specs[s][a];
                for (int d = 0; d < nl; ++d) {
                    const uint64_t n = S.send_cnt[rank_base_ + d];
                    if (!n) continue;
                    const XferSpec& D = specs[d][a];
                    if (D.recv_cnt[rank_base_ + s] != n)
                        throw_fmt(RJ_ERR_DEVICE, "exchange counts disagree (%d -> %d)", s, d);
                    uint8_t*       dst = D.recv + D.recv_off[rank_base_ + s];
                    const uint8_t* src = S.send + S.send_off[rank_base_ + d];
                    if (lanes_[s]->device == lanes_[d]->device)
                        RJ_HIP(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, xfer_[s]));
                    else
                        RJ_HIP(hipMemcpyPeerAsync(dst, lanes_[d]->device, src, lanes_[s]->device, n, xfer_[s]));
                }
            }
            RJ_HIP(hipEventRecord(sent_[s], xfer_[s]));
        }
        for (int l = 0; l < nl; ++l) {
            RJ_HIP(hipSetDevice(lanes_[l]->device));
            for (int s = 0; s < nl; ++s) RJ_HIP(hipStreamWaitEvent(xfer_[l], sent_[s], 0));
            if (t1[l]) RJ_HIP(hipEventRecord(t1[l], xfer_[l]));
            RJ_HIP(hipEventRecord(done[l], xfer_[l]));
        }
    } else {
        Rccl& R = rccl();
        for (int l = 0; l < nl; ++l) {
            RJ_HIP(hipSetDevice(lanes_[l]->device));
            RJ_HIP(hipStreamWaitEvent(xfer_[l], ready[l], 0));
            if (t0[l]) RJ_HIP(hipEventRecord(t0[l], xfer_[l]));
        }
        // slices travel in pieces of at most 1 GiB (both ends cut the same total the same
        // way): multi-GiB point-to-point operations are outside what RCCL is exercised with
        constexpr uint64_t PIECE = 1ull << 30;
        // The helper thread gets the operations BY VALUE: should the wait below expire and the helper
        // have to be given up inside RCCL, it must not wake up later to references into this frame.
        struct Op {
            int         dev;
            ncclComm_t  comm;
            hipStream_t stream;
            uint8_t*    ptr;
            uint64_t    bytes;
            int         peer;
            bool        send;
        };
        auto ops = std::make_shared<std::vector<Op>>();
        for (int l = 0; l < nl; ++l) {
            ncclComm_t c = static_cast<ncclComm_t>(nccl_[l]);
            for (const XferSpec& X : specs[l])
                for (int p = 0; p < world_; ++p) {
                    for (uint64_t o = 0; o < X.send_cnt[p]; o += PIECE)
                        ops->push_back(Op{lanes_[l]->device, c, xfer_[l], const_cast<uint8_t*>(X.send) + X.send_off[p] + o,
                                          std::min(PIECE, X.send_cnt[p] - o), p, true});
                    for (uint64_t o = 0; o < X.recv_cnt[p]; o += PIECE)
                        ops->push_back(Op{lanes_[l]->device, c, xfer_[l], X.recv + X.recv_off[p] + o,
                                          std::min(PIECE, X.recv_cnt[p] - o), p, false});
                }
        }
        Rccl* const rp = &R;  // (process lifetime)
        bounded(name, [ops, rp] {
            RJ_NCCL(rp->GroupStart());
            int dev = -1;
            for (const Op& op : *ops) {
                if (op.dev != dev) RJ_HIP(hipSetDevice(dev = op.dev));
                if (op.send)
                    RJ_NCCL(rp->Send(op.ptr, op.bytes, ncclUint8, op.peer, op.comm, op.stream));
                else
                    RJ_NCCL(rp->Recv(op.ptr, op.bytes, ncclUint8, op.peer, op.comm, op.stream));
            }
            RJ_NCCL(rp->GroupEnd());
        });
        for (int l = 0; l < nl; ++l) {
            RJ_HIP(hipSetDevice(lanes_[l]->device));
            if (t1[l]) RJ_HIP(hipEventRecord(t1[l], xfer_[l]));
            RJ_HIP(hipEventRecord(done[l], xfer_[l]));
        }
    }
    RJ_HIP(hipSetDevice(lanes_[0]->device));
}

}  // namespace rj
