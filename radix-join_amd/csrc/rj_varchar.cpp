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
//                            ColumnInserter<string> (reference include/plan.h:301-334),
//                            in parallel slabs.
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

struct Lookup {
    const uint8_t* const*        pages;
    uint64_t                     n_pages;
    const std::vector<uint64_t>& row_base;
    std::string                  scratch;  // long strings are stitched here

    // -> false for NULL; otherwise [*p, *p + *len) holds the string until the next call
    bool get(uint64_t row, const char** p, uint32_t* len) {
        if (row >= row_base[n_pages]) return false;  // rows the pages do not cover are NULL
        uint64_t pg = (uint64_t)(std::upper_bound(row_base.begin(), row_base.end(), row) -
                                 row_base.begin()) - 1;
        const uint8_t* page = pages[pg];
        uint16_t       nr = rd16(page);
        if (nr == 0xffff) {
            scratch.assign(reinterpret_cast<const char*>(page + 4), rd16(page + 2));
            for (uint64_t q = pg + 1; q < n_pages && rd16(pages[q]) == 0xfffe; ++q)
                scratch.append(reinterpret_cast<const char*>(pages[q] + 4), rd16(pages[q] + 2));
            *p = scratch.data();
            *len = (uint32_t)scratch.size();
            return true;
        }
        uint32_t       i = (uint32_t)(row - row_base[pg]);
        const uint8_t* bitmap = page + PAGE_BYTES - (nr + 7) / 8;
        if (!((bitmap[i >> 3] >> (i & 7)) & 1)) return false;
        uint32_t       idx = popcount_below(bitmap, i);
        uint16_t       nnn = rd16(page + 2);
        const uint8_t* offs = page + 4;
        const char*    data = reinterpret_cast<const char*>(page) + 4 + (size_t)nnn * 2;
        uint32_t       end = rd16(offs + (size_t)idx * 2);
        uint32_t       beg = idx ? rd16(offs + (size_t)(idx - 1) * 2) : 0;
        *p = data + beg;
        *len = end - beg;
        return true;
    }
};

struct PageWriter {
    std::vector<uint8_t>& out;
    uint64_t&             n_pages;
    uint16_t              num_rows = 0;
    std::vector<uint16_t> offs;
    std::vector<char>     chars;
    std::vector<uint8_t>  bitmap;

    uint8_t* new_page() {
        out.resize(out.size() + PAGE_BYTES, 0);
        ++n_pages;
        return out.data() + out.size() - PAGE_BYTES;
    }
    void bit(uint16_t idx, bool set) {
        while (bitmap.size() < (size_t)idx / 8 + 1) bitmap.push_back(0);
        if (set) bitmap[idx / 8] |= (uint8_t)(1u << (idx % 8));
    }
    void save_page() {
        uint8_t* page = new_page();
        uint16_t nv = (uint16_t)offs.size();
        memcpy(page, &num_rows, 2);
        memcpy(page + 2, &nv, 2);
        if (nv) memcpy(page + 4, offs.data(), (size_t)nv * 2);
        if (!chars.empty()) memcpy(page + 4 + (size_t)nv * 2, chars.data(), chars.size());
        memcpy(page + PAGE_BYTES - bitmap.size(), bitmap.data(), bitmap.size());
        num_rows = 0;
        offs.clear();
        chars.clear();
        bitmap.clear();
    }
    void save_long(const char* s, size_t len) {
        size_t off = 0;
        bool   first = true;
        while (off < len) {
            uint8_t* page = new_page();
            uint16_t tag = first ? 0xffff : 0xfffe;
            first = false;
            size_t   chunk = std::min<size_t>(len - off, PAGE_BYTES - 4);
            uint16_t n16 = (uint16_t)chunk;
            memcpy(page, &tag, 2);
            memcpy(page + 2, &n16, 2);
            memcpy(page + 4, s + off, chunk);
            off += chunk;
        }
    }
    void add_null() {
        if (4 + offs.size() * 2 + chars.size() + (num_rows / 8 + 1) > PAGE_BYTES) save_page();
        bit(num_rows, false);
        ++num_rows;
    }
    void add(const char* p, uint32_t len) {
        if (len > PAGE_BYTES - 7) {
            if (num_rows > 0) save_page();
            save_long(p, len);
            return;
        }
        if (4 + (offs.size() + 1) * 2 + (chars.size() + len) + (num_rows / 8 + 1) > PAGE_BYTES)
            save_page();
        bit(num_rows, true);
        chars.insert(chars.end(), p, p + len);
        offs.push_back((uint16_t)chars.size());
        ++num_rows;
    }
    void finish() {
        if (num_rows) save_page();
    }
};
}  // namespace

void varchar_gather_encode(const uint8_t* const* pages, uint64_t n_pages,
                           const std::vector<uint64_t>& row_base, const uint32_t* idx, uint64_t n,
                           std::vector<uint8_t>& out_pages, uint64_t& n_out_pages) {
    // Slabs of 64 K rows are encoded independently (each starts a fresh page — a valid,
    // marginally less dense layout than one greedy pass) so large outputs use all host cores.
    const uint64_t SLAB = 1u << 16;
    uint64_t       n_slabs = (n + SLAB - 1) / SLAB;
    out_pages.clear();
    n_out_pages = 0;
    // The lookups are a chain of dependent cache misses into random pages (header -> bitmap
    // -> offset array -> characters).  Rows are resolved in batches, one link of the chain at
    // a time with the next link prefetched, so the misses of a batch overlap.
    auto encode = [&](uint64_t b, uint64_t e, std::vector<uint8_t>& out, uint64_t& np) {
        Lookup     lk{pages, n_pages, row_base, {}};
        PageWriter w{out, np};
        out.reserve((e - b) * 24 + PAGE_BYTES);
        constexpr int  B = 32;
        const uint8_t* page[B];
        const uint8_t* bitmap[B];
        const char*    str[B];
        uint32_t       at[B], len[B];
        int            kind[B];  // 0 = NULL, 1 = string in str/len, 2 = long string (slow path)
        const uint64_t covered = row_base[n_pages];
        for (uint64_t base = b; base < e; base += B) {
            const int m = (int)std::min<uint64_t>(B, e - base);
            for (int k = 0; k < m; ++k) {
                uint64_t row = idx[base + k];
                page[k] = nullptr;
                if (row >= covered) continue;  // rows the pages do not cover are NULL
                uint64_t pg = (uint64_t)(std::upper_bound(row_base.begin(), row_base.end(), row) -
                                         row_base.begin()) - 1;
                page[k] = pages[pg];
                at[k] = (uint32_t)(row - row_base[pg]);
                __builtin_prefetch(page[k]);
            }
            for (int k = 0; k < m; ++k) {
                kind[k] = 0;
                if (!page[k]) continue;
                uint16_t nr = rd16(page[k]);
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
                if (kind[k] != 1) continue;
                const uint8_t* offs = page[k] + 4;
                uint16_t       nnn = rd16(page[k] + 2);
                uint32_t       end = rd16(offs + (size_t)at[k] * 2);
                uint32_t       beg = at[k] ? rd16(offs + (size_t)(at[k] - 1) * 2) : 0;
                str[k] = reinterpret_cast<const char*>(page[k]) + 4 + (size_t)nnn * 2 + beg;
                len[k] = end - beg;
                __builtin_prefetch(str[k]);
                __builtin_prefetch(str[k] + len[k]);
            }
            for (int k = 0; k < m; ++k) {
                if (kind[k] == 1) {
                    w.add(str[k], len[k]);
                } else if (kind[k] == 2) {
                    const char* p;
                    uint32_t    l;
                    if (lk.get(idx[base + k], &p, &l))
                        w.add(p, l);
                    else
                        w.add_null();
                } else {
                    w.add_null();
                }
            }
        }
        w.finish();
    };
    if (n_slabs <= 1) {
        encode(0, n, out_pages, n_out_pages);
        return;
    }
    std::vector<std::vector<uint8_t>> parts(n_slabs);
    std::vector<uint64_t>             counts(n_slabs, 0);
    parallel_for(n_slabs, 1, [&](size_t b, size_t e) {
        for (size_t s = b; s < e; ++s)
            encode(s * SLAB, std::min<uint64_t>(n, (s + 1) * SLAB), parts[s], counts[s]);
    });
    size_t total = 0;
    for (auto& p : parts) total += p.size();
    out_pages.resize(total);
    size_t off = 0;
    for (uint64_t s = 0; s < n_slabs; ++s) {
        memcpy(out_pages.data() + off, parts[s].data(), parts[s].size());
        off += parts[s].size();
        n_out_pages += counts[s];
    }
}

}  // namespace rj
