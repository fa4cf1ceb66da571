// rj_hostpool.cpp — persistent host worker threads.
//
// The host side of the path is memcpy-shaped work in 8 KiB units: gathering the caller's
// individually allocated Pages into pinned staging, scattering result pages back, encoding
// VARCHAR slabs.  Spawning threads per 32 MiB chunk cost as much as the copy itself for the
// small JOB tables, so the workers are started once per process and parked on a condition
// variable between jobs.
#include <atomic>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <thread>

#include "rj_internal.hpp"

namespace rj {
namespace {

struct Job {
    size_t                                      n, grain;
    const std::function<void(size_t, size_t)>*  fn;
    std::atomic<size_t>                         next{0};
    int                                         active = 0;  // workers inside the job (pool mutex)
    std::exception_ptr                          error;       // first failure (pool mutex)
};

class HostPool {
   public:
    HostPool() {
        unsigned hw = std::thread::hardware_concurrency();
        unsigned nt = std::min<unsigned>(hw ? hw : 4, 16);
        for (unsigned i = 1; i < nt; ++i) std::thread([this] { worker(); }).detach();
        n_workers_ = nt > 0 ? nt - 1 : 0;
    }

    void run(size_t n, size_t grain, const std::function<void(size_t, size_t)>& fn) {
        if (n == 0) return;
        if (grain == 0) grain = 1;
        if (n <= grain || n_workers_ == 0) {
            fn(0, n);
            return;
        }
        std::lock_guard<std::mutex> one(run_m_);  // one job at a time
        Job job;
        job.n = n;
        job.grain = grain;
        job.fn = &fn;
        {
            std::lock_guard<std::mutex> g(m_);
            cur_ = &job;
            ++gen_;
        }
        work_cv_.notify_all();
        drain(job);
        std::unique_lock<std::mutex> g(m_);
        cur_ = nullptr;  // nobody attaches any more
        done_cv_.wait(g, [&] { return job.active == 0; });
        if (job.error) std::rethrow_exception(job.error);
    }

   private:
    void drain(Job& job) {
        try {
            for (;;) {
                size_t b = job.next.fetch_add(job.grain);
                if (b >= job.n) break;
                (*job.fn)(b, std::min(job.n, b + job.grain));
            }
        } catch (...) {
            job.next.store(job.n);  // stop handing out work
            std::lock_guard<std::mutex> g(m_);
            if (!job.error) job.error = std::current_exception();
        }
    }

    void worker() {
        uint64_t seen = 0;
        for (;;) {
            Job* job = nullptr;
            {
                std::unique_lock<std::mutex> g(m_);
                work_cv_.wait(g, [&] { return gen_ != seen; });
                seen = gen_;
                job = cur_;
                if (job) ++job->active;
            }
            if (!job) continue;
            drain(*job);
            {
                std::lock_guard<std::mutex> g(m_);
                --job->active;
            }
            done_cv_.notify_all();
        }
    }

    std::mutex              run_m_, m_;
    std::condition_variable work_cv_, done_cv_;
    Job*                    cur_ = nullptr;
    uint64_t                gen_ = 0;
    unsigned                n_workers_ = 0;
};

}  // namespace

void parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t)>& fn) {
    // never destroyed: the parked workers must not outlive their mutexes at process exit
    static HostPool* pool = new HostPool();
    pool->run(n, grain, fn);
}

}  // namespace rj
