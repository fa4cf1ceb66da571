// rj_comm.hip — exchange transports of the sharded join (see rj_comm.hpp).
#include "rj_comm.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library itself is dlopen'ed (no link-time dependency)

#include <algorithm>
#include <mutex>

#include "rj_internal.hpp"

namespace rj {

namespace {

struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId)    GetUniqueId = nullptr;
    decltype(&ncclCommInitRank)   CommInitRank = nullptr;
    decltype(&ncclCommDestroy)    CommDestroy = nullptr;
    decltype(&ncclGroupStart)     GroupStart = nullptr;
    decltype(&ncclGroupEnd)       GroupEnd = nullptr;
    decltype(&ncclSend)           Send = nullptr;
    decltype(&ncclRecv)           Recv = nullptr;
    decltype(&ncclAllGather)      AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl& rccl() {
    static Rccl       r;
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    if (r.h) return r;
    // the soname first: a process that already holds a librccl (PyTorch bundles one) binds to it
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void*       h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) throw_fmt(RJ_ERR_DEVICE, "cannot load librccl (%s): multi-process sharding needs RCCL", dlerror());
    auto sym = [&](const char* n) {
        void* p = dlsym(h, n);
        if (!p) throw_fmt(RJ_ERR_DEVICE, "librccl lacks %s", n);
        return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.h = h;
    return r;
}

#define RJ_NCCL(expr)                                                                          \
    do {                                                                                       \
        ncclResult_t _r = (expr);                                                              \
        if (_r != ncclSuccess)                                                                 \
            ::rj::throw_fmt(RJ_ERR_DEVICE, "%s failed: %s", #expr, rccl().GetErrorString(_r)); \
    } while (0)

}  // namespace

void Comm::make_id(rj_comm_id* out) {
    static_assert(sizeof(ncclUniqueId) == RJ_COMM_ID_BYTES, "rj_comm_id is an ncclUniqueId");
    ncclUniqueId id;
    RJ_NCCL(rccl().GetUniqueId(&id));
    memcpy(out->bytes, id.internal, RJ_COMM_ID_BYTES);
}

Comm::Comm(std::vector<Context*> lanes, int world, int rank_base, int mode, const rj_comm_id* id)
    : lanes_(std::move(lanes)), world_(world), rank_base_(rank_base) {
    const int nl = (int)lanes_.size();
    if (nl < 1 || world < nl || rank_base < 0 || rank_base + nl > world)
        throw_fmt(RJ_ERR_ARG, "bad rank layout: %d local ranks from %d in a world of %d", nl, rank_base, world);
    if (world & (world - 1)) throw_fmt(RJ_ERR_UNSUPPORTED, "world size must be a power of two");
    const bool all_local = nl == world;
    if (mode == 0) mode = all_local ? P2P : RCCL;
    if (mode == P2P && !all_local)
        throw_fmt(RJ_ERR_ARG, "peer-to-peer exchange needs every rank in this process (world %d, local %d)", world, nl);
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
        if (nl > 1) RJ_NCCL(R.GroupStart());
        for (int l = 0; l < nl; ++l) {
            RJ_HIP(hipSetDevice(lanes_[l]->device));
            ncclComm_t c = nullptr;
            RJ_NCCL(R.CommInitRank(&c, world_, nid, rank_base_ + l));
            nccl_[l] = c;
        }
        if (nl > 1) RJ_NCCL(R.GroupEnd());
    }
    RJ_HIP(hipSetDevice(lanes_[0]->device));
}

Comm::~Comm() {
    for (size_t l = 0; l < lanes_.size(); ++l) {
        (void)hipSetDevice(lanes_[l]->device);
        if (xfer_[l]) (void)hipStreamSynchronize(xfer_[l]);
        if (nccl_[l]) (void)rccl().CommDestroy(static_cast<ncclComm_t>(nccl_[l]));
        if (cnt_dev_[l]) (void)hipFree(cnt_dev_[l]);
        if (sent_[l]) (void)hipEventDestroy(sent_[l]);
        if (xfer_[l]) (void)hipStreamDestroy(xfer_[l]);
    }
    if (!lanes_.empty()) (void)hipSetDevice(lanes_[0]->device);
}

void Comm::allgather_u64(const std::vector<std::vector<uint64_t>>& vals, size_t k,
                         std::vector<std::vector<uint64_t>>& all) {
    const int nl = n_local();
    all.assign(world_, std::vector<uint64_t>(k, 0));
    if (mode_ == P2P) {
        for (int l = 0; l < nl; ++l) all[rank_base_ + l] = vals[l];
        return;
    }
    Rccl& R = rccl();
    const size_t need = (size_t)(world_ + 1) * k * 8;
    if (need > cnt_cap_) {
        for (int l = 0; l < nl; ++l) {
            RJ_HIP(hipSetDevice(lanes_[l]->device));
            if (cnt_dev_[l]) RJ_HIP(hipFree(cnt_dev_[l]));
            cnt_dev_[l] = nullptr;
            RJ_HIP(hipMalloc(&cnt_dev_[l], need));
        }
        cnt_cap_ = need;
    }
    for (int l = 0; l < nl; ++l) {
        RJ_HIP(hipSetDevice(lanes_[l]->device));
        RJ_HIP(hipMemcpyAsync(cnt_dev_[l], vals[l].data(), k * 8, hipMemcpyHostToDevice, xfer_[l]));
    }
    if (nl > 1) RJ_NCCL(R.GroupStart());
    for (int l = 0; l < nl; ++l) {
        RJ_HIP(hipSetDevice(lanes_[l]->device));
        uint64_t* base = static_cast<uint64_t*>(cnt_dev_[l]);
        RJ_NCCL(R.AllGather(base, base + k, k, ncclUint64, static_cast<ncclComm_t>(nccl_[l]), xfer_[l]));
    }
    if (nl > 1) RJ_NCCL(R.GroupEnd());
    std::vector<uint64_t> host((size_t)world_ * k);
    for (int l = 0; l < nl; ++l) {
        RJ_HIP(hipSetDevice(lanes_[l]->device));
        if (l == 0)
            RJ_HIP(hipMemcpyAsync(host.data(), static_cast<uint64_t*>(cnt_dev_[l]) + k, (size_t)world_ * k * 8,
                                  hipMemcpyDeviceToHost, xfer_[l]));
        RJ_HIP(hipStreamSynchronize(xfer_[l]));
    }
    for (int r = 0; r < world_; ++r) all[r].assign(host.begin() + (size_t)r * k, host.begin() + (size_t)(r + 1) * k);
    RJ_HIP(hipSetDevice(lanes_[0]->device));
}

void Comm::all_to_all(const std::vector<std::vector<XferSpec>>& specs, const std::vector<hipEvent_t>& ready,
                      std::vector<hipEvent_t>& done, const char* name) {
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
        RJ_NCCL(R.GroupStart());
        for (int l = 0; l < nl; ++l) {
            RJ_HIP(hipSetDevice(lanes_[l]->device));
            ncclComm_t c = static_cast<ncclComm_t>(nccl_[l]);
            for (const XferSpec& X : specs[l])
                for (int p = 0; p < world_; ++p) {
                    for (uint64_t o = 0; o < X.send_cnt[p]; o += PIECE)
                        RJ_NCCL(R.Send(X.send + X.send_off[p] + o, std::min(PIECE, X.send_cnt[p] - o), ncclUint8, p, c,
                                       xfer_[l]));
                    for (uint64_t o = 0; o < X.recv_cnt[p]; o += PIECE)
                        RJ_NCCL(R.Recv(X.recv + X.recv_off[p] + o, std::min(PIECE, X.recv_cnt[p] - o), ncclUint8, p, c,
                                       xfer_[l]));
                }
        }
        RJ_NCCL(R.GroupEnd());
        for (int l = 0; l < nl; ++l) {
            RJ_HIP(hipSetDevice(lanes_[l]->device));
            if (t1[l]) RJ_HIP(hipEventRecord(t1[l], xfer_[l]));
            RJ_HIP(hipEventRecord(done[l], xfer_[l]));
        }
    }
    RJ_HIP(hipSetDevice(lanes_[0]->device));
}

}  // namespace rj
