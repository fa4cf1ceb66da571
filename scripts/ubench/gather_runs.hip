// Micro-benchmark: what would a join pay for reading its partitions as GATHERED runs?
// Today the last radix pass writes every final partition contiguously (after a histogram pass over
// the segment) and the join reads ~45 KB per partition in one piece.  If the last pass sorted each
// 16384-tuple tile by the second digit IN PLACE instead (sequential writes, no histogram pass), a
// final partition (segment s, digit d) would be the union, over the segment's ~120 tiles, of one
// run of ~32 tuples each — at arbitrary 12-byte offsets inside tiles 192 KiB apart.  This measures
// the read side of that trade with stand-in data: workgroup w reads partition (s, d) either as one
// contiguous piece, or as T runs of R bytes at stride TILE, the workgroups of a segment taking
// consecutive digits (so that the lines two neighbouring runs share are wanted by neighbouring
// workgroups at about the same time).
// build: hipcc --offload-arch=gfx950 -O3 -o gather_runs gather_runs.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

constexpr uint32_t TUPLE = 12, TILE_TUPLES = 16384, F = 512;
constexpr uint64_t TILE_BYTES = (uint64_t)TILE_TUPLES * TUPLE;  // 192 KiB

// contiguous: partition p = bytes [p * part_bytes, (p + 1) * part_bytes), read as dwords
__global__ __launch_bounds__(512) void k_contig(const uint32_t* in, uint32_t part_dwords, uint32_t n_part, uint32_t* sink) {
    uint32_t acc = 0;
    for (uint32_t p = blockIdx.x; p < n_part; p += gridDim.x) {
        const uint32_t* b = in + (size_t)p * part_dwords;
        for (uint32_t i = threadIdx.x; i < part_dwords; i += 512) acc ^= b[i];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// gathered: partition (s, d) = for every tile t of segment s the dwords of run d inside the tile
// (run d of a tile = tuples [d * TILE_TUPLES / F + jitter(t, d) ...), here simply equal runs with a
// per-tile shift so that they do not start on line boundaries); a wave takes a run at a time
__global__ __launch_bounds__(512) void k_gather(const uint32_t* in, uint32_t tiles_per_seg, uint32_t n_seg, uint32_t* sink) {
    const uint32_t run_dwords = TILE_TUPLES / F * TUPLE / 4;  // 96 dwords = 384 bytes
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    uint32_t       acc = 0;
    for (uint32_t p = blockIdx.x; p < n_seg * F; p += gridDim.x) {
        const uint32_t s = p / F, d = p % F;
        const uint32_t* seg = in + (size_t)s * tiles_per_seg * (TILE_BYTES / 4);
        for (uint32_t t = wid; t < tiles_per_seg; t += 8) {
            // runs shift by 3 tuples per tile inside their slot: arbitrary 12-byte alignment
            const uint32_t shift = (t * 7u + 3u) % (TILE_TUPLES / F);
            const uint32_t first = (d * (TILE_TUPLES / F) + shift) % TILE_TUPLES * (TUPLE / 4);
            const uint32_t* r = seg + (size_t)t * (TILE_BYTES / 4) + first;
            for (uint32_t i = lane; i < run_dwords; i += 64) acc ^= r[i < (TILE_BYTES / 4 - first) ? i : 0];
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const uint32_t n_seg = argc > 1 ? atoi(argv[1]) : 512;
    const uint32_t tiles_per_seg = 120;
    const size_t   total = (size_t)n_seg * tiles_per_seg * TILE_BYTES;  // 11.8 GB
    uint32_t *     in, *sink;
    CK(hipMalloc(&in, total));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(in, 1, total));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const uint32_t n_part = n_seg * F, part_dwords = (uint32_t)(total / 4 / n_part);
    printf("%u segments x %u tiles x 192 KiB = %.1f GB; %u partitions of %.1f KB, gathered as %u runs of 384 bytes\n", n_seg,
           tiles_per_seg, total / 1e9, n_part, part_dwords * 4 / 1e3, tiles_per_seg);
    for (int rep = 0; rep < 3; ++rep) {
        for (int grid : {512, 2048, 8192, 65536}) {
            float ms;
            CK(hipEventRecord(a));
            k_contig<<<grid, 512>>>(in, part_dwords, n_part, sink);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b));
            const float c = ms;
            CK(hipEventRecord(a));
            k_gather<<<grid, 512>>>(in, tiles_per_seg, n_seg, sink);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b));
            if (rep)
                printf("grid %6d: contiguous %.2f ms = %.0f GB/s   gathered runs %.2f ms = %.0f GB/s\n", grid, c, total / c / 1e6, ms,
                       total / ms / 1e6);
        }
    }
    return 0;
}
