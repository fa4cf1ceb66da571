// rj_comm.hpp — the one exchange step of a sharded join: a variable-size all-to-all between the
// ranks of a job (one rank per GPU), plus the tiny all-gathers of counts / status words around it.
//
// Two transports behind one interface:
//   P2P   every rank lives in THIS process (one context owning N devices): slices move with
//         hipMemcpyPeerAsync, one copy per (source, destination) pair, all pairs in flight at
//         once — on MI355X every pair of GPUs has its own xGMI link, so the all-to-all is not
//         ring-bound;
//   RCCL  ranks live in several processes (one process per GPU, e.g. bench.py under torchrun):
//         grouped ncclSend/ncclRecv on a communicator created from an rj_comm_id (= ncclUniqueId)
//         the host program passed to every rank.  librccl is dlopen'ed on first use, so a
//         single-GPU deployment never loads it (and a process that already holds PyTorch's copy
//         binds to that one through the shared soname).
//
// Nothing here waits without a bound.  Host calls into RCCL that can block on a peer (communicator
// bring-up, the connection set-up inside the first ncclGroupEnd towards a peer) run on a helper
// thread the caller waits for against a deadline; streams and events are polled against the same
// deadline.  When it passes, the communicators are aborted (ncclCommAbort), the Comm is marked
// failed — every later call throws at once — and RJ_ERR_DEVICE is raised: the job is lost, the
// process should exit non-zero (a fresh process may retry).
// There is no reference counterpart (the reference is one CPU process).
#pragma once
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rj.h"

namespace rj {

struct Context;

// One local rank's half of an all-to-all of bytes.
struct XferSpec {
    const uint8_t*        send = nullptr;  // device memory of the local rank
    std::vector<uint64_t> send_off, send_cnt;  // [world] bytes, per destination rank
    uint8_t*              recv = nullptr;  // device memory of the local rank
    std::vector<uint64_t> recv_off, recv_cnt;  // [world] bytes, per source rank
};

class Comm {
   public:
    enum Mode { P2P = 1, RCCL = 2 };
    // lanes: the contexts of the local ranks (rank_base + i); world = ranks of the whole job;
    // timeout_ms: the bound on every wait; bringup_ms (0 = the same): the bound on communicator
    // bring-up alone, which in a cold process (RCCL loading its kernels) can take seconds
    Comm(std::vector<Context*> lanes, int world, int rank_base, int mode, const rj_comm_id* id, int timeout_ms,
         int bringup_ms = 0);
    ~Comm();
    int  world() const { return world_; }
    int  rank_base() const { return rank_base_; }
    int  n_local() const { return (int)lanes_.size(); }
    Mode mode() const { return mode_; }
    int  timeout_ms() const { return timeout_ms_; }
    bool failed() const { return failed_; }

    // vals[l][0..k) of every local rank l  ->  all[r][0..k) for every rank r of the job.
    // Synchronises the host with the exchange streams (bounded).
    void allgather_u64(const std::vector<std::vector<uint64_t>>& vals, size_t k,
                       std::vector<std::vector<uint64_t>>& all);

    // One all-to-all of a relation: specs[l] = local rank l's slices, one XferSpec per array of
    // the tuple layout (one for packed pairs, two for key array + carry-pair array); all arrays
    // travel in ONE group.  Enqueued on the exchange streams, which first wait for `ready[l]`
    // (recorded on lane l's compute stream once its send buffers are complete).  On return
    // done[l] has been recorded: lane l's recv buffers are complete when it fires — wait for it
    // with wait_event(), never with an unbounded host wait.  Buffers must stay alive until then.
    // `name`: what a profiling context books the exchange under (time on the exchange stream of
    // each local rank from "my send buffers are ready" to "everything of mine has arrived / left")
    void all_to_all(const std::vector<std::vector<XferSpec>>& specs, const std::vector<hipEvent_t>& ready,
                    std::vector<hipEvent_t>& done, const char* name);

    // Bounded host waits on lane `lane`'s device: poll until the event / the exchange stream has
    // completed or the deadline passes (then: abort, mark failed, throw RJ_ERR_DEVICE).
    void wait_event(int lane, hipEvent_t ev, const char* what);
    void wait_stream(int lane, const char* what);

    hipStream_t xfer_stream(int lane) const { return xfer_[lane]; }

    static void make_id(rj_comm_id* out);  // ncclGetUniqueId

   private:
    std::vector<Context*>    lanes_;
    int                      world_ = 1, rank_base_ = 0;
    Mode                     mode_ = P2P;
    int                      timeout_ms_ = 120000;
    bool                     failed_ = false;
    std::string              failed_what_;
    std::vector<hipStream_t> xfer_;
    std::vector<void*>       nccl_;       // ncclComm_t per local rank (RCCL mode)
    std::vector<void*>       cnt_dev_;    // per lane: small device buffer for count all-gathers
    void*                    cnt_host_ = nullptr;  // pinned: lane 0's gathered counts land here
    std::vector<hipEvent_t>  sent_;       // P2P: per lane, "all my outgoing copies are enqueued and done"
    size_t                   cnt_cap_ = 0;

    // helper thread for host calls that may block on a peer
    struct Worker;
    std::shared_ptr<Worker> worker_;
    void bounded(const char* what, std::function<void()> fn, int ms = 0);
    void mark_failed(const char* what, const char* why);  // aborts the communicators; does not throw
    [[noreturn]] void fail(const char* what, const char* why, int ms = 0);
    void check_alive() const;
};

}  // namespace rj
