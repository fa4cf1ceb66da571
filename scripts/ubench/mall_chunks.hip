// Micro-benchmark: can a later radix pass read its input from HBM only ONCE?
// The pass needs two sweeps over every input segment (histogram, then scatter).  Run segment-group
// by segment-group — histogram launch over a chunk, then scatter launch over the same chunk — the
// second sweep could be served by the 256 MiB Infinity Cache instead of HBM, if a chunk (plus what
// streams through the cache between its two uses) stays resident.  This measures exactly that with
// stand-in kernels: k_read (the histogram's sweep) and k_copy (the scatter: read chunk, write chunk),
// over chunk sizes 8 MiB .. 1 GiB, on one stream and on two interleaved streams, against the
// un-chunked baseline (read everything, then copy everything).
// build: hipcc --offload-arch=gfx950 -O3 -o mall_chunks mall_chunks.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// every thread reads 4 x 16 bytes per iteration, grid-strided over [0, n_vec)
__global__ __launch_bounds__(1024) void k_read(const u32x4* p, size_t n_vec, uint32_t* sink) {
    uint32_t     acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
        const u32x4 a = p[i];
        acc += a[0] ^ a[1] ^ a[2] ^ a[3];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(1024) void k_copy(const u32x4* p, u32x4* q, size_t n_vec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) q[i] = p[i];
}

int main(int argc, char** argv) {
    const size_t total = (size_t)(argc > 1 ? atoll(argv[1]) : 6144) << 20;  // bytes per buffer
    u32x4 *      in, *out;
    uint32_t*    sink;
    CK(hipMalloc(&in, total));
    CK(hipMalloc(&out, total));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(in, 1, total));
    CK(hipMemset(out, 2, total));
    hipStream_t st[2];
    CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
    hipEvent_t a, b, j;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
    const size_t nv = total / 16;
    auto         grid_for = [](size_t n_vec) {  // one 16 KiB tile per workgroup pass, at most 2048 workgroups
        size_t g = (n_vec + 1023) / 1024;
        return (unsigned)(g > 2048 ? 2048 : (g ? g : 1));
    };
    auto timed = [&](const char* what, auto&& body, double algo_bytes) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a, st[0]));
            body();
            // join stream 1 into stream 0
            CK(hipEventRecord(j, st[1]));
            CK(hipStreamWaitEvent(st[0], j, 0));
            CK(hipEventRecord(b, st[0]));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        printf("%-64s %8.3f ms  %7.0f GB/s algorithmic\n", what, best, algo_bytes / best / 1e6);
        fflush(stdout);
        return best;
    };
    const double B = (double)total;
    printf("buffers: %zu MiB in, %zu MiB out\n", total >> 20, total >> 20);
    timed("read everything (one launch)", [&] { k_read<<<grid_for(nv), 1024, 0, st[0]>>>(in, nv, sink); }, B);
    timed("copy everything (one launch)", [&] { k_copy<<<grid_for(nv), 1024, 0, st[0]>>>(in, out, nv); }, 2 * B);
    timed("read everything, then copy everything (two launches)",
          [&] {
              k_read<<<grid_for(nv), 1024, 0, st[0]>>>(in, nv, sink);
              k_copy<<<grid_for(nv), 1024, 0, st[0]>>>(in, out, nv);
          },
          3 * B);
    for (size_t chunk_mb : {8, 16, 32, 48, 64, 96, 128, 192, 256, 512, 1024}) {
        const size_t cv = (chunk_mb << 20) / 16;
        for (int streams = 1; streams <= 2; ++streams) {
            char what[128];
            snprintf(what, sizeof what, "chunks of %4zu MiB: read chunk, copy chunk; %d stream%s", chunk_mb, streams,
                     streams > 1 ? "s" : "");
            timed(what,
                  [&] {
                      size_t k = 0;
                      for (size_t o = 0; o < nv; o += cv, ++k) {
                          const size_t n = nv - o < cv ? nv - o : cv;
                          hipStream_t  s = st[streams > 1 ? k & 1 : 0];
                          k_read<<<grid_for(n), 1024, 0, s>>>(in + o, n, sink);
                          k_copy<<<grid_for(n), 1024, 0, s>>>(in + o, out + o, n);
                      }
                  },
                  3 * B);
        }
    }
    // the same with the copy alone per chunk (launch overhead of chunking without the re-read)
    for (size_t chunk_mb : {16, 64, 256}) {
        const size_t cv = (chunk_mb << 20) / 16;
        char         what[128];
        snprintf(what, sizeof what, "chunks of %4zu MiB: copy chunk only; 1 stream", chunk_mb);
        timed(what,
              [&] {
                  for (size_t o = 0; o < nv; o += cv) {
                      const size_t n = nv - o < cv ? nv - o : cv;
                      k_copy<<<grid_for(n), 1024, 0, st[0]>>>(in + o, out + o, n);
                  }
              },
              2 * B);
    }
    return 0;
}
