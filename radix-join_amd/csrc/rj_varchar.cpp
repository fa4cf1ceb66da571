// rj_varchar.cpp — host side of VARCHAR late materialisation.
//
// VARCHAR is never a join key in the workload (reference ANNOUNCEMENTS.md:11), so
// strings never travel to the GPU: a VARCHAR column moves through the plan as a
// row-id column of its base table and is resolved here, at the root.
//   varchar_dir_build      — a page directory (rows before each page): 4 bytes read per
//                            page, instead of decoding every row the way the VARCHAR branch
//                            of Table::from_columnar does (reference
//                            src/build_table.cpp:382-428)
//   varchar_gather_encode  — row ids -> strings (page lookup + bitmap popcount + offset
//                            array, the layout of build_table.cpp:406-427 / long-string
//                            pages :384-405) -> pages with the fill rule of
//                            Table::to_columnar (reference src/build_table.cpp:595-677) /
//                            ColumnInserter<string> (reference include/plan.h:301-334):
//                            lookup (parallel) -> page boundaries (the fill rule as counter
//                            arithmetic) -> every page written in place (parallel), the
//                            shape of the device encoder in rj_varchar_dev.hip.
#include <algorithm>
#include <cstring>

#include "rj_internal.hpp"

namespace rj {

static inline uint16_t rd16(const uint8_t* p) {
    uint16_t v;
    memcpy(&v, p, 2);
    return v;
}

void varchar_dir_build(const uint8_t* const* pages, uint64_t n_pages, uint64_t num_rows,
                       std::vector<uint64_t>& row_base) {
    row_base.assign(n_pages + 1, 0);
    uint64_t rows = 0;
    for (uint64_t p = 0; p < n_pages; ++p) {
        row_base[p] = rows;
        uint16_t nr = rd16(pages[p]);
        if (nr == 0xffff)
            rows += 1;  // first page of a long string: one row (build_table.cpp:384-391)
        else if (nr == 0xfffe) {
            if (p == 0 || rows == 0)
                throw_fmt(RJ_ERR_DATA, "long string page 0xfffe must follows a string");
        } else
            rows += nr;
    }
    row_base[n_pages] = rows;
    // more rows in the pages than the table declares: the reference throws "row_idx" when a
    // NON-NULL string lands at a row index >= num_rows (:388,:419); NULL rows past the end
    // are tolerated.  Rare: walk the pages that reach past the end.
    if (rows > num_rows) {
        for (uint64_t p = 0; p < n_pages; ++p) {
            uint16_t nr = rd16(pages[p]);
            if (nr == 0xfffe) continue;
            if (nr == 0xffff) {
                if (row_base[p] >= num_rows) throw_fmt(RJ_ERR_DATA, "row_idx");
                continue;
            }
            if (row_base[p] + nr <= num_rows) continue;
            const uint8_t* bm = pages[p] + PAGE_BYTES - (nr + 7) / 8;
            for (uint32_t i = 0; i < nr; ++i)
                if (row_base[p] + i >= num_rows && ((bm[i >> 3] >> (i & 7)) & 1))
                    throw_fmt(RJ_ERR_DATA, "row_idx");
        }
    }
}

namespace {

// bits set in bitmap[0, i)
inline uint32_t popcount_below(const uint8_t* bitmap, uint32_t i) {
    uint32_t c = 0, w = 0;
    for (; w + 64 <= i; w += 64) {
        uint64_t x;
        memcpy(&x, bitmap + w / 8, 8);
        c += (uint32_t)__builtin_popcountll(x);
    }
    for (; w + 8 <= i; w += 8) c += (uint32_t)__builtin_popcount(bitmap[w / 8]);
    if (w < i) c += (uint32_t)__builtin_popcount(bitmap[w / 8] & ((1u << (i - w)) - 1u));
    return c;
}

// One result row after the lookup: where its characters sit.  Short strings point into the source
// page; a long string (0xffff / 0xfffe page chain) names the first page of its chain.
struct StrRef {
    const char* p = nullptr;
    uint32_t    len = 0;        // characters; NULL_LEN = NULL row
    uint32_t    chain = 0;      // long string: index of its 0xffff page + 1, else 0
};
constexpr uint32_t NULL_LEN = 0xffffffffu;

// One output page, decided by the walk before a single byte is written (the host twin of
// k_vc_walk / k_vc_encode, csrc/rj_varchar_dev.hip): a normal page holds result rows
// [first, first + nr); piece k of a long string holds up to PAGE_BYTES - 4 of its characters.
struct OutPage {
    uint64_t first = 0;
    uint32_t nr = 0;
    uint32_t piece = 0;  // 0 = normal page, 1 + k = piece k of row `first`
};

// The fill rule of the reference's encoders (Table::to_columnar, src/build_table.cpp:595-677;
// ColumnInserter<std::string>, include/plan.h:301-334) as arithmetic over three counters — rows,
// non-NULL values and characters of the page being filled: a row goes to the next page when
//     4 + 2 * values + chars + (rows / 8 + 1) > PAGE_BYTES
// would hold with it in; a string longer than PAGE_BYTES - 7 closes the page and takes pages of
// its own.
void walk_pages(const std::vector<StrRef>& refs, std::vector<OutPage>& pages) {
    uint64_t first = 0;
    uint32_t rows = 0, vals = 0, chars = 0;
    auto     close = [&](uint64_t next_first) {
        if (rows) pages.push_back(OutPage{first, rows, 0});
        first = next_first;
        rows = vals = chars = 0;
    };
    for (uint64_t r = 0; r < refs.size(); ++r) {
        const StrRef& s = refs[r];
        if (s.len != NULL_LEN && s.len > PAGE_BYTES - 7) {
            close(r + 1);
            for (uint32_t off = 0, k = 0; off < s.len; off += PAGE_BYTES - 4, ++k)
                pages.push_back(OutPage{r, 0, 1 + k});
            continue;
        }
        const uint32_t add_v = s.len == NULL_LEN ? 0u : 1u, add_c = s.len == NULL_LEN ? 0u : s.len;
        if (4 + 2 * (vals + add_v) + (chars + add_c) + (rows / 8 + 1) > PAGE_BYTES) close(r);
        ++rows;
        vals += add_v;
        chars += add_c;
    }
    close(refs.size());
}

// characters [off, off + n) of the long string whose chain starts at source page `pg`
void copy_chain(const uint8_t* const* pages, uint64_t n_pages, uint64_t pg, uint32_t off, uint32_t n, uint8_t* dst) {
    uint32_t at = 0;
    for (uint64_t q = pg; q < n_pages && n; ++q) {
        if (q != pg && rd16(pages[q]) != 0xfffe) break;
        const uint32_t have = rd16(pages[q] + 2);
        if (off < at + have) {
            const uint32_t b = off - at, take = std::min(n, have - b);
            memcpy(dst, pages[q] + 4 + b, take);
            dst += take;
            off += take;
            n -= take;
        }
        at += have;
    }
}

void write_page(const uint8_t* const* src_pages, uint64_t n_src, const std::vector<StrRef>& refs, const OutPage& op,
                uint8_t* page) {
    memset(page, 0, PAGE_BYTES);
    if (op.piece) {  // piece of a long string: tag, character count, characters
        const StrRef&  s = refs[op.first];
        const uint32_t off = (op.piece - 1) * (PAGE_BYTES - 4);
        const uint16_t tag = op.piece == 1 ? 0xffff : 0xfffe, n16 = (uint16_t)std::min<uint32_t>(s.len - off, PAGE_BYTES - 4);
        memcpy(page, &tag, 2);
        memcpy(page + 2, &n16, 2);
        copy_chain(src_pages, n_src, s.chain - 1, off, n16, page + 4);
        return;
    }
    uint16_t nv = 0;
    for (uint32_t i = 0; i < op.nr; ++i) nv += refs[op.first + i].len != NULL_LEN;
    const uint16_t nr16 = (uint16_t)op.nr;
    memcpy(page, &nr16, 2);
    memcpy(page + 2, &nv, 2);
    uint8_t* offs = page + 4;
    uint8_t* text = page + 4 + (size_t)nv * 2;
    uint8_t* bitmap = page + PAGE_BYTES - (op.nr + 7) / 8;
    uint16_t end = 0, v = 0;
    for (uint32_t i = 0; i < op.nr; ++i) {
        const StrRef& s = refs[op.first + i];
        if (s.len == NULL_LEN) continue;
        bitmap[i >> 3] |= (uint8_t)(1u << (i & 7));
        if (s.len) memcpy(text + end, s.p, s.len);
        end = (uint16_t)(end + s.len);
        memcpy(offs + (size_t)v * 2, &end, 2);
        ++v;
    }
}
}  // namespace

void varchar_gather_encode(const uint8_t* const* pages, uint64_t n_pages,
                           const std::vector<uint64_t>& row_base, const uint32_t* idx, uint64_t n,
                           std::vector<uint8_t>& out_pages, uint64_t& n_out_pages) {
    out_pages.clear();
    n_out_pages = 0;
    // ---- 1. where does every result row's string sit?  The lookups are a chain of dependent
    // cache misses into random pages (header -> bitmap -> offset array); rows are resolved in
    // batches, one link of the chain at a time with the next link prefetched, so the misses of
    // a batch overlap.  Parallel over the rows.
    std::vector<StrRef> refs(n);
    const uint64_t      covered = row_base[n_pages];
    parallel_for(n, 1u << 14, [&](size_t b, size_t e) {
        constexpr int  B = 32;
        const uint8_t* page[B];
        const uint8_t* bitmap[B];
        uint64_t       pgi[B];
        uint32_t       at[B];
        int            kind[B];  // 0 = NULL, 1 = short string, 2 = long string
        for (uint64_t base = b; base < e; base += B) {
            const int m = (int)std::min<uint64_t>(B, e - base);
            for (int k = 0; k < m; ++k) {
                const uint64_t row = idx[base + k];
                page[k] = nullptr;
                if (row >= covered) continue;  // rows the pages do not cover are NULL
                pgi[k] = (uint64_t)(std::upper_bound(row_base.begin(), row_base.end(), row) - row_base.begin()) - 1;
                page[k] = pages[pgi[k]];
                at[k] = (uint32_t)(row - row_base[pgi[k]]);
                __builtin_prefetch(page[k]);
            }
            for (int k = 0; k < m; ++k) {
                kind[k] = 0;
                if (!page[k]) continue;
                const uint16_t nr = rd16(page[k]);
                if (nr == 0xffff) {
                    kind[k] = 2;
                    continue;
                }
                bitmap[k] = page[k] + PAGE_BYTES - (nr + 7) / 8;
                __builtin_prefetch(bitmap[k]);
                __builtin_prefetch(bitmap[k] + (at[k] >> 3));
                kind[k] = 1;
            }
            for (int k = 0; k < m; ++k) {
                if (kind[k] != 1) continue;
                if (!((bitmap[k][at[k] >> 3] >> (at[k] & 7)) & 1)) {
                    kind[k] = 0;
                    continue;
                }
                at[k] = popcount_below(bitmap[k], at[k]);  // index among the non-NULL values
                __builtin_prefetch(page[k] + 4 + (size_t)at[k] * 2);
            }
            for (int k = 0; k < m; ++k) {
                StrRef& r = refs[base + k];
                if (kind[k] == 0) {
                    r.len = NULL_LEN;
                } else if (kind[k] == 1) {
                    const uint8_t* offs = page[k] + 4;
                    const uint16_t nnn = rd16(page[k] + 2);
                    const uint32_t end = rd16(offs + (size_t)at[k] * 2);
                    const uint32_t beg = at[k] ? rd16(offs + (size_t)(at[k] - 1) * 2) : 0;
                    r.p = reinterpret_cast<const char*>(page[k]) + 4 + (size_t)nnn * 2 + beg;
                    r.len = end - beg;
                    __builtin_prefetch(r.p);
                } else {  // a long string: its length is the sum over its page chain
                    uint32_t len = rd16(page[k] + 2);
                    for (uint64_t q = pgi[k] + 1; q < n_pages && rd16(pages[q]) == 0xfffe; ++q) len += rd16(pages[q] + 2);
                    r.len = len;
                    r.chain = (uint32_t)pgi[k] + 1;
                    if (len <= PAGE_BYTES - 7) {
                        // (a chain the reference's encoders would never have produced: short enough
                        // for a normal page.  Materialise it so that it can be copied like one.)
                        char* tmp = new char[len ? len : 1];
                        copy_chain(pages, n_pages, pgi[k], 0, len, reinterpret_cast<uint8_t*>(tmp));
                        r.p = tmp;
                    }
                }
            }
        }
    });
    // ---- 2. page boundaries: the sequential fill rule, counters only
    std::vector<OutPage> plan;
    plan.reserve(n / 64 + 16);
    walk_pages(refs, plan);
    // ---- 3. every page is written straight into its final place, pages in parallel
    n_out_pages = plan.size();
    out_pages.resize(plan.size() * (size_t)PAGE_BYTES);
    uint8_t* out = out_pages.data();
    parallel_for(plan.size(), 16, [&](size_t b, size_t e) {
        for (size_t p = b; p < e; ++p) write_page(pages, n_pages, refs, plan[p], out + p * PAGE_BYTES);
    });
    for (StrRef& r : refs)
        if (r.chain && r.len <= PAGE_BYTES - 7) delete[] r.p;
}

}  // namespace rj
