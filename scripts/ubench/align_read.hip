// Micro-benchmark: does a dword-aligned (not 16-byte aligned) global_load_dwordx4 stream read
// slower than a 16-byte aligned one?  (Page images start their values at byte 4.)
// build: hipcc --offload-arch=gfx950 -O3 -o align_read align_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));

__global__ __launch_bounds__(1024) void k_read(const uint8_t* p, size_t n_vec, uint32_t off, uint32_t* out) {
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t       i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // 4 independent loads in flight per thread per iteration
    for (; i + 3 * stride < n_vec; i += 4 * stride) {
        u32x4a a = *reinterpret_cast<const u32x4a*>(p + off + i * 16);
        u32x4a b = *reinterpret_cast<const u32x4a*>(p + off + (i + stride) * 16);
        u32x4a c = *reinterpret_cast<const u32x4a*>(p + off + (i + 2 * stride) * 16);
        u32x4a d = *reinterpret_cast<const u32x4a*>(p + off + (i + 3 * stride) * 16);
        acc += a[0] ^ a[1] ^ a[2] ^ a[3] ^ b[0] ^ b[1] ^ b[2] ^ b[3] ^ c[0] ^ c[1] ^ c[2] ^ c[3] ^ d[0] ^ d[1] ^ d[2] ^ d[3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    uint8_t*     p;
    uint32_t*    out;
    hipMalloc(&p, bytes + 64);
    hipMalloc(&out, 4);
    hipMemset(p, 1, bytes + 64);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep)
        for (uint32_t off : {0u, 4u, 8u, 0u, 4u}) {
            hipEventRecord(a);
            for (int it = 0; it < 10; ++it) k_read<<<2048, 1024>>>(p, bytes / 16, off, out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            printf("offset %u: %.0f GB/s\n", off, 10.0 * bytes / ms / 1e6);
        }
    return 0;
}
