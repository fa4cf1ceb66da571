// rj_varchar_dev.hip — VARCHAR late materialisation on the device.
//
// A VARCHAR column travels through a plan as a row-id column of its base table; at the root
// the strings of the result rows are gathered from the base column's pages and encoded as
// VARCHAR pages.  For small results the host does that (rj_varchar.cpp); for large ones the
// gather + encode below run on the GPU and only finished pages cross PCIe:
//   k_vc_resolve   row id -> {source page, byte offset, length} | NULL | long string
//                  (page lookup by binary search in the row directory, bitmap test + popcount
//                  for the value index, offset array: the decode of reference
//                  src/build_table.cpp:382-428 for ONE row)
//   k_vc_walk      the page-fill rule of Table::to_columnar (reference src/build_table.cpp:
//                  595-677: a row goes to the current page while header + offsets + chars +
//                  bitmap fit in 8192 bytes, strings above 8185 bytes become 0xffff/0xfffe page
//                  chains) is sequential, so it runs sequentially — one LANE per chunk of
//                  VC_CHUNK rows, every chunk starting a fresh page (a valid, marginally less
//                  dense layout: Table::from_columnar decodes any page sequence).  Pass 1 counts
//                  the chunk's pages, pass 2 (after a scan of the counts) records each page's
//                  first row, row count and kind.
//   k_vc_encode    one workgroup per output page: header, end-offset array (block scan of the
//                  lengths), characters, validity bitmap (wave ballots).
#include <hip/hip_ext.h>

#include "rj_kernels.hpp"

namespace rj {

namespace {

__device__ __forceinline__ uint32_t rd16(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

constexpr uint32_t VC_NULL = 0xffffffffu;        // VcRow::len of a NULL row
constexpr uint32_t VC_LONG = 0xffffffffu;        // VcRow::beg of a long string (len = total characters)
constexpr uint32_t VC_MAX_INLINE = PAGE_BYTES - 7;  // longest string a normal page holds (:644)
constexpr uint32_t VC_LONG_CHUNK = PAGE_BYTES - 4;  // characters per long-string page (:614)

// bits set in bitmap[0, i)
__device__ __forceinline__ uint32_t popcount_below(const uint8_t* bitmap, uint32_t i) {
    uint32_t c = 0, w = 0;
    for (; w + 8 <= i; w += 8) c += (uint32_t)__popc((uint32_t)bitmap[w >> 3]);
    if (w < i) c += (uint32_t)__popc((uint32_t)bitmap[w >> 3] & ((1u << (i - w)) - 1u));
    return c;
}

}  // namespace

// row id -> where its string sits
__device__ __forceinline__ VcRow vc_resolve_row(const uint8_t* pages, uint32_t n_pages, const uint32_t* row_base,
                                                uint32_t row) {
    VcRow r{0, 0, VC_NULL};
    if (row >= row_base[n_pages]) return r;  // rows the pages do not cover are NULL
    // largest pg with row_base[pg] <= row (pages holding no row, 0xfffe, share their
    // successor's base: the search lands behind them)
    uint32_t lo = 0, hi = n_pages;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (row_base[mid] <= row)
            lo = mid;
        else
            hi = mid;
    }
    const uint8_t* page = pages + (size_t)lo * PAGE_BYTES;
    const uint32_t nr = rd16(page);
    if (nr == 0xffffu) {  // long string: this page + the 0xfffe pages behind it
        uint32_t total = rd16(page + 2);
        for (uint32_t q = lo + 1; q < n_pages && rd16(pages + (size_t)q * PAGE_BYTES) == 0xfffeu; ++q)
            total += rd16(pages + (size_t)q * PAGE_BYTES + 2);
        return VcRow{lo, VC_LONG, total};
    }
    const uint32_t at = row - row_base[lo];
    const uint8_t* bitmap = page + PAGE_BYTES - (nr + 7) / 8;
    if ((bitmap[at >> 3] >> (at & 7u)) & 1u) {
        const uint32_t idx = popcount_below(bitmap, at);  // index among the non-NULL values
        const uint32_t nnn = rd16(page + 2);
        const uint32_t end = rd16(page + 4 + (size_t)idx * 2);
        const uint32_t beg = idx ? rd16(page + 4 + (size_t)(idx - 1) * 2) : 0u;
        r = VcRow{lo, 4u + nnn * 2u + beg, end - beg};
    }
    return r;
}

__global__ __launch_bounds__(256) void k_vc_resolve(const uint8_t* pages, uint32_t n_pages,
                                                    const uint32_t* row_base, const uint32_t* rowids,
                                                    uint32_t n, VcRow* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = vc_resolve_row(pages, n_pages, row_base, rowids[i]);
}

// Byte `pos` of a resolved string (long strings: walk the page chain — rare, slow, correct).
__device__ __forceinline__ uint32_t vc_byte(const uint8_t* pages, uint32_t n_pages, const VcRow& r, uint32_t pos) {
    if (r.beg != VC_LONG) return pages[(size_t)r.page * PAGE_BYTES + r.beg + pos];
    uint32_t q = r.page;
    while (q < n_pages) {
        const uint8_t* page = pages + (size_t)q * PAGE_BYTES;
        const uint32_t nc = rd16(page + 2);
        if (pos < nc) return page[4 + pos];
        pos -= nc;
        ++q;
    }
    return 0;
}

// ---- VARCHAR join keys (reference hash_join_omp<std::string>, src/execute.cpp:33-38,278):
// the strings of a key column are hashed to 64 bits (FNV-1a, as the reference does) and joined as
// 64-bit keys; every matching pair is then compared byte for byte (k_vc_verify), so hash
// collisions cannot add rows.  rows[i] keeps where row i's string sits for that comparison.
__global__ __launch_bounds__(256) void k_vc_hash(const uint8_t* pages, uint32_t n_pages, const uint32_t* row_base,
                                                 const uint32_t* rowids /* nullptr: row i itself */, uint32_t n,
                                                 VcRow* rows, uint64_t* hash, uint8_t* valid, uint64_t hash_mask) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const VcRow r = vc_resolve_row(pages, n_pages, row_base, rowids ? rowids[i] : i);
    rows[i] = r;
    uint64_t h = 0xcbf29ce484222325ull;
    if (r.len != VC_NULL) {
        if (r.beg != VC_LONG) {
            const uint8_t* s = pages + (size_t)r.page * PAGE_BYTES + r.beg;
            for (uint32_t k = 0; k < r.len; ++k) h = (h ^ s[k]) * 0x100000001b3ull;
        } else {
            uint32_t q = r.page, left = r.len;
            while (left && q < n_pages) {
                const uint8_t* page = pages + (size_t)q * PAGE_BYTES;
                const uint32_t nc = min(left, rd16(page + 2));
                for (uint32_t k = 0; k < nc; ++k) h = (h ^ page[4 + k]) * 0x100000001b3ull;
                left -= nc;
                ++q;
            }
        }
    }
    hash[i] = h & hash_mask;  // (all ones; tests narrow it to force collisions)
    valid[i] = r.len != VC_NULL;
}

// keep[i] = the build and the probe string of output pair i are equal; *n_bad counts the others
__global__ __launch_bounds__(256) void k_vc_verify(const uint8_t* pages_b, uint32_t np_b, const VcRow* rows_b,
                                                   const uint8_t* pages_p, uint32_t np_p, const VcRow* rows_p,
                                                   const uint32_t* bidx, const uint32_t* pidx, uint32_t n,
                                                   uint8_t* keep, unsigned long long* n_bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const VcRow a = rows_b[bidx[i]], b = rows_p[pidx[i]];
    bool        eq = a.len == b.len && a.len != VC_NULL;
    if (eq) {
        if (a.beg != VC_LONG && b.beg != VC_LONG) {
            const uint8_t* x = pages_b + (size_t)a.page * PAGE_BYTES + a.beg;
            const uint8_t* y = pages_p + (size_t)b.page * PAGE_BYTES + b.beg;
            for (uint32_t k = 0; k < a.len && eq; ++k) eq = x[k] == y[k];
        } else {
            for (uint32_t k = 0; k < a.len && eq; ++k)
                eq = vc_byte(pages_b, np_b, a, k) == vc_byte(pages_p, np_p, b, k);
        }
    }
    keep[i] = eq;
    if (!eq) atomicAdd(n_bad, 1ull);
}

// order-free compaction of the surviving pairs (only runs when a hash collision was found)
__global__ __launch_bounds__(256) void k_vc_compact(const uint8_t* keep, const uint32_t* bidx, const uint32_t* pidx,
                                                    uint32_t n, uint32_t* out_b, uint32_t* out_p,
                                                    unsigned long long* cursor) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool     k = i < n && keep[i];
    const uint64_t mask = __ballot(k);
    if (mask == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(cursor, (unsigned long long)__popcll(mask));
    base = __shfl(base, 0);
    if (k) {
        const uint32_t o = (uint32_t)base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                                      __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        out_b[o] = bidx[i];
        out_p[o] = pidx[i];
    }
}

// The page-fill rule, one lane per chunk.  WRITE = false: count the chunk's pages.
// WRITE = true: record every page of the chunk at page_base[chunk] + k.
template <bool WRITE>
__global__ __launch_bounds__(64) void k_vc_walk(const VcRow* rows, uint32_t n, uint32_t* pages_in_chunk,
                                                const uint32_t* page_base, VcPage* page_out) {
    const uint32_t chunk = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = chunk * VC_CHUNK;
    if (b >= n) return;
    const uint32_t e = min(n, b + VC_CHUNK);
    uint32_t       np = 0;                      // pages closed so far
    uint32_t       nr = 0, nv = 0, chars = 0;   // the open page
    uint32_t       first = b;
    VcPage*        po = WRITE ? page_out + page_base[chunk] : nullptr;
    auto close = [&](uint32_t next_first) {
        if (nr) {
            if (WRITE) po[np] = VcPage{first, nr, 0u};
            ++np;
        }
        nr = nv = chars = 0;
        first = next_first;
    };
    for (uint32_t i = b; i < e; ++i) {
        const VcRow r = rows[i];
        if (r.len == VC_NULL) {
            if (4u + nv * 2u + chars + (nr / 8u + 1u) > PAGE_BYTES) close(i);
            ++nr;
        } else if (r.beg == VC_LONG && r.len > VC_MAX_INLINE) {
            close(i);
            const uint32_t k = (r.len + VC_LONG_CHUNK - 1) / VC_LONG_CHUNK;
            if (WRITE)
                for (uint32_t s = 0; s < k; ++s) po[np + s] = VcPage{i, 0u, 1u + s};  // kind: 1 + piece
            np += k;
            first = i + 1;
        } else {
            // (a source long string no longer than 8185 characters cannot exist: the encoder
            // only makes page chains for longer ones; it would take this branch and be copied
            // piece by piece all the same)
            if (4u + (nv + 1u) * 2u + (chars + r.len) + (nr / 8u + 1u) > PAGE_BYTES) close(i);
            ++nr;
            ++nv;
            chars += r.len;
        }
    }
    close(e);
    if (!WRITE) pages_in_chunk[chunk] = np;
}

// Copy `len` characters of a source string that may be a long-string page chain starting at
// source page `pg` (skipping `skip` characters), cooperatively by the calling threads.
__device__ __forceinline__ void copy_chars(uint8_t* dst, const uint8_t* pages, uint32_t n_pages,
                                           const VcRow& r, uint32_t skip, uint32_t len, uint32_t tid,
                                           uint32_t nthreads) {
    if (r.beg != VC_LONG) {
        const uint8_t* src = pages + (size_t)r.page * PAGE_BYTES + r.beg + skip;
        for (uint32_t k = tid; k < len; k += nthreads) dst[k] = src[k];
        return;
    }
    // walk the chain: page q holds rd16(+2) characters from byte 4
    uint32_t q = r.page, done = 0;
    while (len && q < n_pages) {
        const uint8_t* page = pages + (size_t)q * PAGE_BYTES;
        const uint32_t nc = rd16(page + 2);
        if (skip >= nc) {
            skip -= nc;
        } else {
            const uint32_t take = min(len, nc - skip);
            for (uint32_t k = tid; k < take; k += nthreads) dst[done + k] = page[4 + skip + k];
            done += take;
            len -= take;
            skip = 0;
        }
        ++q;
    }
}

__global__ __launch_bounds__(256) void k_vc_encode(const uint8_t* pages, uint32_t n_pages, const VcRow* rows,
                                                   const VcPage* plist, uint8_t* out) {
    __shared__ uint32_t s_w[4], s_w2[4];
    const VcPage        pg = plist[blockIdx.x];
    uint8_t*            page = out + (size_t)blockIdx.x * PAGE_BYTES;
    const uint32_t      lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    if (pg.kind != 0) {  // piece pg.kind - 1 of a long string (reference build_table.cpp:603-622)
        const VcRow    r = rows[pg.first];
        const uint32_t piece = pg.kind - 1, skip = piece * VC_LONG_CHUNK;
        const uint32_t nc = min(VC_LONG_CHUNK, r.len - skip);
        if (threadIdx.x == 0) {
            const uint32_t tag = piece == 0 ? 0xffffu : 0xfffeu;
            page[0] = (uint8_t)tag;
            page[1] = (uint8_t)(tag >> 8);
            page[2] = (uint8_t)nc;
            page[3] = (uint8_t)(nc >> 8);
        }
        copy_chars(page + 4, pages, n_pages, r, skip, nc, threadIdx.x, 256);
        // the rest of a result page is unspecified (the host path leaves zeros there too)
        for (uint32_t k = 4 + nc + threadIdx.x; k < PAGE_BYTES; k += 256) page[k] = 0;
        return;
    }
    // ---- a normal page of pg.nr rows: count the non-NULL ones first (the characters start
    //      behind the offset array)
    uint32_t cnt = 0;
    for (uint32_t i = threadIdx.x; i < pg.nr; i += 256) cnt += rows[pg.first + i].len != VC_NULL;
    for (int off = 32; off; off >>= 1) cnt += __shfl_down(cnt, off);
    if (lane == 0) s_w[wid] = cnt;
    __syncthreads();
    const uint32_t nv = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    const uint32_t nb = (pg.nr + 7) / 8;
    uint8_t*       chars = page + 4 + (size_t)nv * 2;
    uint8_t*       bm = page + PAGE_BYTES - nb;
    if (threadIdx.x == 0) {
        page[0] = (uint8_t)pg.nr;
        page[1] = (uint8_t)(pg.nr >> 8);
        page[2] = (uint8_t)nv;
        page[3] = (uint8_t)(nv >> 8);
    }
    uint32_t run_v = 0, run_c = 0;  // non-NULL rows / characters before this slab of 256 rows
    for (uint32_t base = 0; base < pg.nr; base += 256) {
        const uint32_t i = base + threadIdx.x;
        VcRow          r{0, 0, VC_NULL};
        if (i < pg.nr) r = rows[pg.first + i];
        const bool     valid = r.len != VC_NULL;
        const uint32_t len = valid ? r.len : 0u;
        // wave-level inclusive scans of (valid, len), then across the four waves
        const uint64_t mask = __ballot(valid);
        const uint32_t vpre = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        uint32_t       incl = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off);
            if (lane >= (uint32_t)off) incl += t;
        }
        if (lane == 63) {
            s_w[wid] = (uint32_t)__popcll(mask);
            s_w2[wid] = incl;
        }
        __syncthreads();
        uint32_t vbase = run_v, cbase = run_c, vtot = 0, ctot = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            if (k < wid) {
                vbase += s_w[k];
                cbase += s_w2[k];
            }
            vtot += s_w[k];
            ctot += s_w2[k];
        }
        if (valid) {
            const uint32_t vi = vbase + vpre, cend = cbase + incl;
            page[4 + (size_t)vi * 2] = (uint8_t)cend;
            page[5 + (size_t)vi * 2] = (uint8_t)(cend >> 8);
            // this thread copies its own string (JOB strings are tens of bytes)
            copy_chars(chars + (cend - len), pages, n_pages, r, 0, len, 0, 1);
        }
        // bitmap bytes of this slab: a wave's ballot holds 8 of them
        if (lane < 8) {
            const uint32_t byte_idx = (base >> 3) + wid * 8u + lane;
            if (byte_idx < nb) bm[byte_idx] = (uint8_t)(mask >> (lane * 8u));
        }
        run_v += vtot;
        run_c += ctot;
        __syncthreads();
    }
    // zero the gap between the characters and the bitmap
    for (uint32_t k = 4 + nv * 2 + run_c + threadIdx.x; k < PAGE_BYTES - nb; k += 256) page[k] = 0;
}

// ---- launchers
#define RJ_VLAUNCH(L, NAME, KERNEL, GRID, BLOCK, ...)                                          \
    do {                                                                                       \
        hipEvent_t _ev0 = nullptr, _ev1 = nullptr;                                             \
        if ((L).timed && (L).timed((L).self, NAME, &_ev0, &_ev1))                              \
            hipExtLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), 0, (L).stream, _ev0, _ev1,  \
                                  0, __VA_ARGS__);                                             \
        else                                                                                   \
            hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(BLOCK), 0, (L).stream, __VA_ARGS__);   \
        hipError_t _le = hipGetLastError();                                                    \
        if (_le != hipSuccess) launch_failed(NAME, hipGetErrorString(_le), false);             \
    } while (0)

void launch_vc_resolve(const Launch& L, const uint8_t* pages, uint32_t n_pages, const uint32_t* row_base,
                       const uint32_t* rowids, uint32_t n, VcRow* out) {
    if (!n) return;
    RJ_VLAUNCH(L, "varchar_resolve", k_vc_resolve, (n + 255) / 256, 256, pages, n_pages, row_base, rowids, n, out);
}

void launch_vc_hash(const Launch& L, const uint8_t* pages, uint32_t n_pages, const uint32_t* row_base,
                    const uint32_t* rowids, uint32_t n, VcRow* rows, uint64_t* hash, uint8_t* valid,
                    uint64_t hash_mask) {
    if (!n) return;
    RJ_VLAUNCH(L, "varchar_hash", k_vc_hash, (n + 255) / 256, 256, pages, n_pages, row_base, rowids, n, rows, hash, valid,
               hash_mask);
}

void launch_vc_verify(const Launch& L, const uint8_t* pages_b, uint32_t np_b, const VcRow* rows_b,
                      const uint8_t* pages_p, uint32_t np_p, const VcRow* rows_p, const uint32_t* bidx,
                      const uint32_t* pidx, uint32_t n, uint8_t* keep, unsigned long long* n_bad) {
    if (!n) return;
    RJ_VLAUNCH(L, "varchar_verify", k_vc_verify, (n + 255) / 256, 256, pages_b, np_b, rows_b, pages_p, np_p, rows_p,
               bidx, pidx, n, keep, n_bad);
}

void launch_vc_compact(const Launch& L, const uint8_t* keep, const uint32_t* bidx, const uint32_t* pidx, uint32_t n,
                       uint32_t* out_b, uint32_t* out_p, unsigned long long* cursor) {
    if (!n) return;
    RJ_VLAUNCH(L, "varchar_compact", k_vc_compact, (n + 255) / 256, 256, keep, bidx, pidx, n, out_b, out_p, cursor);
}

// RJ_CTX_PREWARM: one harmless launch, so that HIP loads this translation unit's code object when
// the context is built (zeroed[0..64) must be zero: a `keep` mask of zeros compacts nothing)
void prewarm_varchar_dev(const Launch& L, uint32_t* zeroed) {
    RJ_VLAUNCH(L, "prewarm", k_vc_compact, 1, 256, reinterpret_cast<const uint8_t*>(zeroed), zeroed, zeroed, 1u, zeroed,
               zeroed, reinterpret_cast<unsigned long long*>(zeroed));
}

void launch_vc_walk(const Launch& L, const VcRow* rows, uint32_t n, uint32_t* pages_in_chunk,
                    const uint32_t* page_base, VcPage* page_out) {
    if (!n) return;
    const uint32_t chunks = (n + VC_CHUNK - 1) / VC_CHUNK;
    if (page_out)
        RJ_VLAUNCH(L, "varchar_walk", (k_vc_walk<true>), (chunks + 63) / 64, 64, rows, n, pages_in_chunk, page_base,
                   page_out);
    else
        RJ_VLAUNCH(L, "varchar_walk", (k_vc_walk<false>), (chunks + 63) / 64, 64, rows, n, pages_in_chunk, page_base,
                   page_out);
}

void launch_vc_encode(const Launch& L, const uint8_t* pages, uint32_t n_pages, const VcRow* rows,
                      const VcPage* plist, uint32_t n_out_pages, uint8_t* out) {
    if (!n_out_pages) return;
    RJ_VLAUNCH(L, "varchar_encode", k_vc_encode, n_out_pages, 256, pages, n_pages, rows, plist, out);
}

}  // namespace rj
