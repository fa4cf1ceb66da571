// Micro-benchmark: which OUTPUT LAYOUT of a 12-byte-tuple radix pass does the memory system
// like best?  Every workgroup reads 16384 tuples (192 KiB, sequential) and writes them as the
// runs a 512-way pass makes (32 tuples per run on average, starting at arbitrary tuple offsets):
//   A  key array (4 B/tuple: 128-byte runs) + carry-pair array (8 B/tuple: 256-byte runs)
//   B  one array of 12-byte tuples (384-byte runs)
//   C  B + a side array of 16-bit next-pass digits (64-byte runs)
// build: hipcc --offload-arch=gfx950 -O3 -o layout_scatter layout_scatter.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr uint32_t TILE = 16384, THREADS = 1024, F = 512, RUN = TILE / F;  // 32 tuples per run

// tile t's run r lands at tuple index (r * n_tiles + t) * RUN + (r % 7): runs of 32 tuples at
// offsets that are not multiples of anything
__device__ __forceinline__ size_t dst_tuple(uint32_t r, uint32_t t, uint32_t n_tiles, uint32_t o) {
    return ((size_t)r * n_tiles + t) * RUN + o + (r % 7u);
}

template <int MODE>
__global__ __launch_bounds__(THREADS) void k(const uint32_t* in, uint32_t* keys, uint2* pairs, uint32_t* aos,
                                             uint16_t* side, uint32_t n_tiles) {
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint32_t* src = in + (size_t)t * TILE * 3;
        uint32_t        w[16][3];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2)
#pragma unroll
            for (int a = 0; a < 3; ++a) w[k2][a] = src[((size_t)k2 * THREADS + threadIdx.x) * 3 + a];
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
            const uint32_t i = k2 * THREADS + threadIdx.x;  // position in the tile's sorted image
            const uint32_t r = i / RUN, o = i % RUN;
            const size_t   d = dst_tuple(r, t, n_tiles, o);
            if (MODE == 0) {
                keys[d] = w[k2][0];
                pairs[d] = make_uint2(w[k2][1], w[k2][2]);
            } else {
                aos[d * 3 + 0] = w[k2][0];
                aos[d * 3 + 1] = w[k2][1];
                aos[d * 3 + 2] = w[k2][2];
                if (MODE == 2) side[d] = (uint16_t)(w[k2][0] >> 9);
            }
        }
    }
}

int main() {
    const size_t   n = (size_t)1 << 29;  // 512 Mi tuples = 6 GiB in
    const uint32_t n_tiles = (uint32_t)(n / TILE);
    uint32_t *in, *keys, *aos;
    uint2*    pairs;
    uint16_t* side;
    hipMalloc(&in, n * 12);
    hipMalloc(&keys, n * 4 + 4096);
    hipMalloc(&pairs, n * 8 + 4096);
    hipMalloc(&aos, n * 12 + 4096);
    hipMalloc(&side, n * 2 + 4096);
    hipMemset(in, 1, n * 12);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const char* names[] = {"A  key array + pair array      ", "B  12-byte tuples              ", "C  12-byte tuples + u16 digits "};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            hipEventRecord(a);
            for (int it = 0; it < 3; ++it) {
                if (mode == 0) k<0><<<2048, THREADS>>>(in, keys, pairs, aos, side, n_tiles);
                if (mode == 1) k<1><<<2048, THREADS>>>(in, keys, pairs, aos, side, n_tiles);
                if (mode == 2) k<2><<<2048, THREADS>>>(in, keys, pairs, aos, side, n_tiles);
            }
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            printf("%s %.2f ms per 512 Mi tuples  (%.0f GB/s of 24 B/tuple)\n", names[mode], ms / 3, 3.0 * 24.0 * n / ms / 1e6);
        }
    return 0;
}
