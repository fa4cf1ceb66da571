// rj_hostpool.cpp — persistent host worker threads.
//
// The host side of the path is memcpy-shaped work in 8 KiB units: gathering the caller's
// individually allocated Pages into pinned staging, scattering result pages back, encoding
// VARCHAR slabs.  Spawning threads per 32 MiB chunk cost as much as the copy itself for the
// small JOB tables, so the workers are started once per process and parked on a condition
// variable between jobs.
#include <immintrin.h>

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

// One 8 KiB page, destination not read first: a plain memcpy into memory that is not in cache
// costs a read-for-ownership of every destination line on top of the source read and the
// write-back; streaming stores skip it, which is a third of the host memory traffic of the
// gather into pinned staging.  (The other direction, pinned staging -> the caller's result
// pages, measured slower with streaming stores and stays a memcpy.)
__attribute__((target("avx2"))) static void copy_page_avx2(void* dst, const void* src) {
    const __m256i* s = static_cast<const __m256i*>(src);
    __m256i*       d = static_cast<__m256i*>(dst);
    for (size_t i = 0; i < PAGE_BYTES / 32; i += 4) {
        __m256i a = _mm256_loadu_si256(s + i), b = _mm256_loadu_si256(s + i + 1);
        __m256i c = _mm256_loadu_si256(s + i + 2), e = _mm256_loadu_si256(s + i + 3);
        _mm256_stream_si256(d + i, a);
        _mm256_stream_si256(d + i + 1, b);
        _mm256_stream_si256(d + i + 2, c);
        _mm256_stream_si256(d + i + 3, e);
    }
}

void copy_page(void* dst, const void* src) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    // streaming stores need a 32-byte aligned destination (pinned staging is; a caller's Page
    // is only guaranteed alignas(8), reference include/plan.h:54)
    if (avx2 && (reinterpret_cast<uintptr_t>(dst) & 31u) == 0)
        copy_page_avx2(dst, src);
    else
        memcpy(dst, src, PAGE_BYTES);
}

void copy_pages_fence() { _mm_sfence(); }

void parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t)>& fn) {
    // never destroyed: the parked workers must not outlive their mutexes at process exit
    static HostPool* pool = new HostPool();
    pool->run(n, grain, fn);
}

}  // namespace rj
