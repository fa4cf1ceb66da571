// rj_varchar.cpp — host side of VARCHAR late materialisation.
//
// VARCHAR is never a join key in the workload (reference ANNOUNCEMENTS.md:11), so
// strings never travel to the GPU: a VARCHAR column moves through the plan as a
// row-id column of its base table and is resolved here, at the root.
//   varchar_index          — one pass over the pages, restating the VARCHAR branch of
//                            Table::from_columnar (reference src/build_table.cpp:382-428)
//   varchar_gather_encode  — rows -> pages with the fill rule of Table::to_columnar
//                            (reference src/build_table.cpp:595-677) / ColumnInserter<string>
//                            (reference include/plan.h:301-334)
#include <algorithm>
#include <cstring>
#include <thread>

#include "rj_internal.hpp"

namespace rj {

static inline uint16_t rd16(const uint8_t* p) {
    uint16_t v;
    memcpy(&v, p, 2);
    return v;
}

void varchar_index(const uint8_t* const* pages, uint64_t n_pages, uint64_t num_rows,
                   std::vector<StrView>& rows, std::vector<std::string>& stitch) {
    rows.assign(num_rows, StrView{nullptr, 0});
    // long strings are stitched into owned std::strings; reserve so pointers stay valid
    uint64_t n_long = 0;
    for (uint64_t p = 0; p < n_pages; ++p)
        if (rd16(pages[p]) == 0xffff) ++n_long;
    stitch.clear();
    stitch.reserve(n_long);
    uint64_t row = 0;
    int64_t  last_long = -1;  // index into stitch of the row being continued
    for (uint64_t p = 0; p < n_pages; ++p) {
        const uint8_t* page = pages[p];
        uint16_t       nr = rd16(page);
        if (nr == 0xffff) {
            uint16_t nchars = rd16(page + 2);
            if (row >= num_rows) throw_fmt(RJ_ERR_DATA, "row_idx");
            stitch.emplace_back(reinterpret_cast<const char*>(page + 4), nchars);
            last_long = (int64_t)stitch.size() - 1;
            rows[row] = StrView{stitch.back().data(), (uint32_t)stitch.back().size()};
            ++row;
        } else if (nr == 0xfffe) {
            uint16_t nchars = rd16(page + 2);
            if (row == 0 || last_long < 0)
                throw_fmt(RJ_ERR_DATA, "long string page 0xfffe must follows a string");
            std::string& s = stitch[(size_t)last_long];
            s.append(reinterpret_cast<const char*>(page + 4), nchars);
            rows[row - 1] = StrView{s.data(), (uint32_t)s.size()};
        } else {
            last_long = -1;
            uint16_t       nnn = rd16(page + 2);
            const uint8_t* offs = page + 4;
            const char*    data = reinterpret_cast<const char*>(page) + 4 + (size_t)nnn * 2;
            const uint8_t* bitmap = page + PAGE_BYTES - (nr + 7) / 8;
            uint32_t       di = 0, prev = 0;
            for (uint32_t i = 0; i < nr; ++i) {
                if (row >= num_rows) throw_fmt(RJ_ERR_DATA, "row_idx");
                if ((bitmap[i >> 3] >> (i & 7)) & 1) {
                    uint32_t end = rd16(offs + (size_t)di * 2);
                    rows[row] = StrView{data + prev, end - prev};
                    prev = end;
                    ++di;
                }
                ++row;
            }
        }
    }
    // a stitched string may have been re-allocated by append(): refresh the views
    // (append can move the buffer; rows[] of earlier long strings stay valid because
    // every std::string owns its own buffer, but the LAST append per string is what
    // counts — recompute all long-string views once)
    {
        uint64_t r = 0;
        size_t   li = 0;
        for (uint64_t p = 0; p < n_pages; ++p) {
            uint16_t nr = rd16(pages[p]);
            if (nr == 0xffff) {
                rows[r] = StrView{stitch[li].data(), (uint32_t)stitch[li].size()};
                ++li;
                ++r;
            } else if (nr != 0xfffe) {
                r += nr;
            }
        }
    }
}

namespace {
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
    void add(const StrView& v) {
        if (v.p == nullptr) {  // NULL
            if (4 + offs.size() * 2 + chars.size() + (num_rows / 8 + 1) > PAGE_BYTES) save_page();
            bit(num_rows, false);
            ++num_rows;
            return;
        }
        if (v.len > PAGE_BYTES - 7) {
            if (num_rows > 0) save_page();
            save_long(v.p, v.len);
            return;
        }
        if (4 + (offs.size() + 1) * 2 + (chars.size() + v.len) + (num_rows / 8 + 1) > PAGE_BYTES)
            save_page();
        bit(num_rows, true);
        chars.insert(chars.end(), v.p, v.p + v.len);
        offs.push_back((uint16_t)chars.size());
        ++num_rows;
    }
    void finish() {
        if (num_rows) save_page();
    }
};
}  // namespace

void varchar_gather_encode(const std::vector<StrView>& rows, const uint32_t* idx, uint64_t n,
                           std::vector<uint8_t>& out_pages, uint64_t& n_pages) {
    // Pages are independent once their first row is known, but that depends on the
    // greedy fill; encode in parallel slabs (each slab starts a fresh page — a valid,
    // slightly less dense layout) when the output is large.
    const uint64_t SLAB = 1u << 20;
    unsigned       hw = std::thread::hardware_concurrency();
    uint64_t       n_slabs = (n + SLAB - 1) / SLAB;
    out_pages.clear();
    n_pages = 0;
    static const StrView null_view{nullptr, 0};
    auto encode = [&](uint64_t b, uint64_t e, std::vector<uint8_t>& out, uint64_t& np) {
        PageWriter w{out, np};
        for (uint64_t i = b; i < e; ++i) {
            uint32_t r = idx[i];
            w.add(r < rows.size() ? rows[r] : null_view);
        }
        w.finish();
    };
    if (n_slabs <= 1 || hw <= 1) {
        encode(0, n, out_pages, n_pages);
        return;
    }
    std::vector<std::vector<uint8_t>> parts(n_slabs);
    std::vector<uint64_t>             counts(n_slabs, 0);
    unsigned                          nt = std::min<uint64_t>(std::min<unsigned>(hw, 16), n_slabs);
    std::vector<std::thread>          th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            for (uint64_t s = t; s < n_slabs; s += nt)
                encode(s * SLAB, std::min(n, (s + 1) * SLAB), parts[s], counts[s]);
        });
    for (auto& x : th) x.join();
    size_t total = 0;
    for (auto& p : parts) total += p.size();
    out_pages.reserve(total);
    for (uint64_t s = 0; s < n_slabs; ++s) {
        out_pages.insert(out_pages.end(), parts[s].begin(), parts[s].end());
        n_pages += counts[s];
    }
}

}  // namespace rj
