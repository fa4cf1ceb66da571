// Micro-benchmark: the memory-system ceiling of a radix scatter as a function of the length of
// the contiguous runs it writes.  Every workgroup streams a 128 KiB tile in (16-byte loads, as
// the partition kernels do) and writes it out as RUN-byte runs, run r of tile t going to
// partition r's region at slot t — the write pattern of a radix pass with fan-out
// 128 KiB / RUN whose tiles append to every partition in turn.  No LDS, no ranking: what is
// left is what HBM + fabric make of reads next to RUN-byte scattered writes.
// build: hipcc --offload-arch=gfx950 -O3 -o run_scatter run_scatter.hip ; run: ./run_scatter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr uint32_t TILE_BYTES = 128 * 1024, THREADS = 1024;

// shift: every partition region starts `shift_r = (r * 8) % 32` bytes off a 32-byte sector, so
// runs begin and end in the middle of sectors their neighbours (other tiles) complete later —
// what a real scatter's runs do (tuple counts per tile and digit are arbitrary).
// xcd != 0: tile t takes slot (t % 8) * (n_tiles / 8) + t / 8 of every partition, so the runs that
// complete each other's partial sectors come from workgroups 8 apart = on the same XCD (blocks are
// dealt round-robin over the 8 XCDs): their partial lines can meet in that XCD's L2.
__global__ __launch_bounds__(THREADS) void k_scatter_u(const uint2* in, uint2* out, uint32_t n_tiles, uint32_t run_bytes,
                                                       uint32_t xcd, uint32_t shift) {
    for (uint32_t t0 = blockIdx.x; t0 < n_tiles; t0 += gridDim.x) {
        const uint32_t t = xcd ? (t0 % 8u) * (n_tiles / 8u) + t0 / 8u : t0;
        const uint2* src = in + (size_t)t0 * (TILE_BYTES / 8);
        uint2        v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = src[k * THREADS + threadIdx.x];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t byte = (k * THREADS + threadIdx.x) * 8u;
            const uint32_t r = byte / run_bytes, o = byte - r * run_bytes;
            const size_t   dst = ((size_t)r * n_tiles + t) * run_bytes + o + (shift == 1 ? (r * 8u) % 32u : shift);  // 1: per-partition 0/8/16/24; else a constant
            out[dst / 8] = v[k];
        }
    }
}

__global__ __launch_bounds__(THREADS) void k_scatter(const uint4* in, uint4* out, uint32_t n_tiles, uint32_t run_bytes) {
    const uint32_t runs_per_tile = TILE_BYTES / run_bytes;
    for (uint32_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint4* src = in + (size_t)t * (TILE_BYTES / 16);
        uint4        v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[k * THREADS + threadIdx.x];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t byte = (k * THREADS + threadIdx.x) * 16u;  // position in the tile's sorted image
            const uint32_t r = byte / run_bytes, o = byte - r * run_bytes;
            // partition r, slot t
            const size_t dst = ((size_t)r * n_tiles + t) * run_bytes + o;
            (void)runs_per_tile;
            out[dst / 16] = v[k];
        }
    }
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)((argc > 1 ? atof(argv[1]) : 8.0) * (1u << 30));
    const uint32_t n_tiles = (uint32_t)(bytes / TILE_BYTES);
    uint4 *in, *out;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes + 4096) != hipSuccess) return 1;
    hipMemset(in, 1, bytes);
    hipMemset(out, 0, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const uint32_t runs[] = {131072, 64, 128, 256, 384 /* not a divisor: skipped */, 512, 1024, 2048, 4096, 16384, 131072};
    for (int grid : {256, 2048})
        for (uint32_t rb : runs) {
            if (TILE_BYTES % rb) continue;
            k_scatter<<<grid, THREADS>>>(in, out, n_tiles, rb);  // warm
            hipEventRecord(a);
            for (int it = 0; it < 3; ++it) k_scatter<<<grid, THREADS>>>(in, out, n_tiles, rb);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            printf("grid %4d  run %6u B (fan-out %5u): %6.0f GB/s read+write, %.2f ms per %.0f GiB\n", grid, rb,
                   TILE_BYTES / rb, 3.0 * 2.0 * bytes / ms / 1e6, ms / 3, bytes / 1073741824.0);
        }
    // 8-byte stores (what the packed scatter issues), sector-aligned vs shifted runs
    for (uint32_t rb : {131072u, 128u, 256u, 512u, 1024u}) {
        k_scatter_u<<<2048, THREADS>>>((const uint2*)in, (uint2*)out, n_tiles, rb, 0, 0);
        hipEventRecord(a);
        for (int it = 0; it < 3; ++it) k_scatter_u<<<2048, THREADS>>>((const uint2*)in, (uint2*)out, n_tiles, rb, 0, 0);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("8-byte stores, runs ALIGNED to the 32-byte sectors, grid 2048, run %6u B: %6.0f GB/s read+write\n", rb,
               3.0 * 2.0 * bytes / ms / 1e6);
    }
    // which alignment do runs need?  every run shifted by the same constant
    for (uint32_t sh : {8u, 16u, 32u, 64u, 96u, 128u})
        for (uint32_t rb : {256u}) {
            k_scatter_u<<<2048, THREADS>>>((const uint2*)in, (uint2*)out, n_tiles, rb, 0, sh);
            hipEventRecord(a);
            for (int it = 0; it < 3; ++it) k_scatter_u<<<2048, THREADS>>>((const uint2*)in, (uint2*)out, n_tiles, rb, 0, sh);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            printf("8-byte stores, every 256-byte run starts %3u bytes off a 256-byte boundary: %6.0f GB/s read+write\n", sh,
                   3.0 * 2.0 * bytes / ms / 1e6);
        }
    for (uint32_t xcd : {0u, 1u})
        for (uint32_t grid : {2048u})
            for (uint32_t rb : {131072u, 128u, 256u, 512u, 1024u}) {
                k_scatter_u<<<grid, THREADS>>>((const uint2*)in, (uint2*)out, n_tiles, rb, xcd, 1);
                hipEventRecord(a);
                for (int it = 0; it < 3; ++it)
                    k_scatter_u<<<grid, THREADS>>>((const uint2*)in, (uint2*)out, n_tiles, rb, xcd, 1);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                printf("8-byte stores, runs off the 32-byte sectors, %s, grid %4u, run %6u B: %6.0f GB/s read+write\n",
                       xcd ? "neighbours on one XCD " : "neighbours on any XCD", grid, rb, 3.0 * 2.0 * bytes / ms / 1e6);
            }
    return 0;
}
