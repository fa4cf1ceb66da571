// contest_unit_tests.cpp — the reference's 8 known-answer join cases
// (reference tests/unit_tests.cpp:10-282), run through the C++ drop-in boundary
// Contest::build_context / execute / destroy_context exactly as the reference's test
// binary calls it, but written against this repository's contract header and with a
// small self-contained row<->page codec (INT32 + VARCHAR) instead of the reference's
// Table class.  Built and run by tests/test_gpu_cpp_shim.py on the GPU box.
#include <plan.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <functional>
#include <optional>
#include <stdexcept>
#include <string>
#include <variant>
#include <vector>

// ------------------------------------------------------------ tiny test frame
struct Case {
    const char*           name;
    std::function<void()> fn;
};
static std::vector<Case>& cases() {
    static std::vector<Case> v;
    return v;
}
struct Reg {
    Reg(const char* n, std::function<void()> f) { cases().push_back({n, std::move(f)}); }
};
#define RJ_CASE(ident, name) \
    static void ident();     \
    static Reg  reg_##ident(name, ident); \
    static void ident()
#define REQUIRE(expr)                                                                     \
    do {                                                                                  \
        if (!(expr)) throw std::runtime_error(std::string("REQUIRE failed: ") + #expr +   \
                                              " (" + __FILE__ + ":" + std::to_string(__LINE__) + ")"); \
    } while (0)

// --------------------------------------------------------------- row <-> pages
// A cell: NULL, INT32 or VARCHAR (all the unit cases need).
using Cell = std::variant<std::monostate, int32_t, std::string>;
using Row = std::vector<Cell>;

static void put16(std::byte* p, uint16_t v) { memcpy(p, &v, 2); }
static uint16_t get16(const std::byte* p) {
    uint16_t v;
    memcpy(&v, p, 2);
    return v;
}

// Page fill rules: reference include/plan.h:204-221 (INT32), :301-328 (VARCHAR)
static ColumnarTable to_columnar(const std::vector<Row>& rows, const std::vector<DataType>& types) {
    ColumnarTable t;
    t.num_rows = rows.size();
    for (size_t c = 0; c < types.size(); ++c) {
        t.columns.emplace_back(types[c]);
        Column&               col = t.columns.back();
        std::vector<uint8_t>  bitmap;
        std::vector<int32_t>  vals;
        std::vector<uint16_t> offs;
        std::string           chars;
        uint16_t              nr = 0;
        auto setbit = [&](bool v) {
            if (bitmap.size() < size_t(nr) / 8 + 1) bitmap.push_back(0);
            if (v) bitmap[nr / 8] |= uint8_t(1u << (nr % 8));
        };
        auto flush = [&] {
            if (nr == 0) return;
            std::byte* p = col.new_page()->data;
            memset(p, 0, PAGE_SIZE);
            put16(p, nr);
            if (types[c] == DataType::INT32) {
                put16(p + 2, uint16_t(vals.size()));
                memcpy(p + 4, vals.data(), vals.size() * 4);
            } else {
                put16(p + 2, uint16_t(offs.size()));
                memcpy(p + 4, offs.data(), offs.size() * 2);
                memcpy(p + 4 + offs.size() * 2, chars.data(), chars.size());
            }
            memcpy(p + PAGE_SIZE - bitmap.size(), bitmap.data(), bitmap.size());
            nr = 0;
            bitmap.clear();
            vals.clear();
            offs.clear();
            chars.clear();
        };
        for (const Row& r : rows) {
            const Cell& v = r[c];
            if (types[c] == DataType::INT32) {
                bool   isv = std::holds_alternative<int32_t>(v);
                size_t need = 4 + (vals.size() + (isv ? 1 : 0)) * 4 + (nr / 8 + 1);
                if (need > PAGE_SIZE) flush();
                setbit(isv);
                if (isv) vals.push_back(std::get<int32_t>(v));
                ++nr;
            } else {
                bool   isv = std::holds_alternative<std::string>(v);
                size_t len = isv ? std::get<std::string>(v).size() : 0;
                size_t need = 4 + (offs.size() + (isv ? 1 : 0)) * 2 + chars.size() + len + (nr / 8 + 1);
                if (need > PAGE_SIZE) flush();
                setbit(isv);
                if (isv) {
                    chars += std::get<std::string>(v);
                    offs.push_back(uint16_t(chars.size()));
                }
                ++nr;
            }
        }
        flush();
    }
    return t;
}

// Page layout: reference src/build_table.cpp:325-343 (INT32), :406-427 (VARCHAR)
static std::vector<Row> from_columnar(const ColumnarTable& t) {
    std::vector<Row> rows(t.num_rows, Row(t.columns.size()));
    for (size_t c = 0; c < t.columns.size(); ++c) {
        size_t row = 0;
        for (const Page* pg : t.columns[c].pages) {
            const std::byte* p = pg->data;
            uint16_t         nr = get16(p), nv = get16(p + 2);
            const auto*      bm = reinterpret_cast<const uint8_t*>(p + PAGE_SIZE - (nr + 7) / 8);
            size_t           di = 0, prev = 0;
            for (uint16_t i = 0; i < nr; ++i, ++row) {
                if (row >= t.num_rows) throw std::runtime_error("row_idx");
                if (!((bm[i / 8] >> (i % 8)) & 1)) continue;
                if (t.columns[c].type == DataType::INT32) {
                    int32_t v;
                    memcpy(&v, p + 4 + di * 4, 4);
                    rows[row][c] = v;
                } else {
                    uint16_t    end = get16(p + 4 + di * 2);
                    const char* data = reinterpret_cast<const char*>(p) + 4 + size_t(nv) * 2;
                    rows[row][c] = std::string(data + prev, data + end);
                    prev = end;
                }
                ++di;
            }
        }
    }
    return rows;
}

static std::vector<Row> run(Plan& plan) {
    void*         ctx = Contest::build_context();
    ColumnarTable res;
    try {
        res = Contest::execute(plan, ctx);
    } catch (...) {
        Contest::destroy_context(ctx);
        throw;
    }
    Contest::destroy_context(ctx);
    auto rows = from_columnar(res);
    std::sort(rows.begin(), rows.end());
    REQUIRE(rows.size() == res.num_rows);
    return rows;
}

static const std::vector<std::tuple<size_t, DataType>> I0{{0, DataType::INT32}};
static const std::vector<std::tuple<size_t, DataType>> I01{{0, DataType::INT32}, {1, DataType::INT32}};

static Plan simple_plan(const std::vector<Row>& a, const std::vector<Row>& b) {
    Plan p;
    p.new_scan_node(0, I0);
    p.new_scan_node(1, I0);
    p.new_join_node(true, 0, 1, 0, 0, I01);
    p.inputs.emplace_back(to_columnar(a, {DataType::INT32}));
    p.inputs.emplace_back(to_columnar(b, {DataType::INT32}));
    p.root = 2;
    return p;
}

static const Cell NUL{};

// ------------------------------------------------------------------- the cases
RJ_CASE(empty_join, "Empty join") {  // unit_tests.cpp:10-28
    Plan p;
    p.new_scan_node(0, I0);
    p.new_scan_node(1, I0);
    p.new_join_node(true, 0, 1, 0, 0, I01);
    ColumnarTable t1, t2;
    t1.columns.emplace_back(DataType::INT32);
    t2.columns.emplace_back(DataType::INT32);
    p.inputs.emplace_back(std::move(t1));
    p.inputs.emplace_back(std::move(t2));
    p.root = 2;
    void* ctx = Contest::build_context();
    auto  res = Contest::execute(p, ctx);
    Contest::destroy_context(ctx);
    REQUIRE(res.num_rows == 0);
    REQUIRE(res.columns.size() == 2);
    REQUIRE(res.columns[0].type == DataType::INT32);
    REQUIRE(res.columns[1].type == DataType::INT32);
}

RJ_CASE(one_line, "One line join") {  // :30-57
    Plan p = simple_plan({{1}}, {{1}});
    REQUIRE(run(p) == (std::vector<Row>{{1, 1}}));
}

RJ_CASE(simple, "Simple join") {  // :59-91
    Plan p = simple_plan({{1}, {2}, {3}}, {{1}, {2}, {3}});
    REQUIRE(run(p) == (std::vector<Row>{{1, 1}, {2, 2}, {3, 3}}));
}

RJ_CASE(empty_result, "Empty Result") {  // :93-123
    Plan p = simple_plan({{1}, {2}, {3}}, {{4}, {5}, {6}});
    void* ctx = Contest::build_context();
    auto  res = Contest::execute(p, ctx);
    Contest::destroy_context(ctx);
    REQUIRE(res.num_rows == 0);
    REQUIRE(res.columns.size() == 2);
    REQUIRE(res.columns[0].type == DataType::INT32);
}

RJ_CASE(same_keys, "Multiple same keys") {  // :125-161
    Plan p = simple_plan({{1}, {1}, {2}, {3}}, {{1}, {1}, {2}, {3}});
    REQUIRE(run(p) == (std::vector<Row>{{1, 1}, {1, 1}, {1, 1}, {1, 1}, {2, 2}, {3, 3}}));
}

RJ_CASE(null_keys, "NULL keys") {  // :163-200
    std::vector<Row> d{{1}, {1}, {NUL}, {2}, {3}};
    Plan             p = simple_plan(d, d);
    REQUIRE(run(p) == (std::vector<Row>{{1, 1}, {1, 1}, {1, 1}, {1, 1}, {2, 2}, {3, 3}}));
}

static Plan multi_col_plan(bool build_left) {  // :202-241, :243-282
    using namespace std::string_literals;
    Plan p;
    p.new_scan_node(0, I0);
    p.new_scan_node(1, {{1, DataType::VARCHAR}, {0, DataType::INT32}});
    p.new_join_node(build_left, 0, 1, 0, 1,
                    {{0, DataType::INT32}, {2, DataType::INT32}, {1, DataType::VARCHAR}});
    std::vector<Row> d{{1, "xxx"s}, {1, "yyy"s}, {NUL, "zzz"s}, {2, "uuu"s}, {3, "vvv"s}};
    p.inputs.emplace_back(to_columnar(d, {DataType::INT32, DataType::VARCHAR}));
    p.inputs.emplace_back(to_columnar(d, {DataType::INT32, DataType::VARCHAR}));
    p.root = 2;
    return p;
}

static std::vector<Row> multi_col_truth() {
    using namespace std::string_literals;
    return {{1, 1, "xxx"s}, {1, 1, "xxx"s}, {1, 1, "yyy"s}, {1, 1, "yyy"s}, {2, 2, "uuu"s}, {3, 3, "vvv"s}};
}

RJ_CASE(multi_cols, "Multiple columns") {
    Plan p = multi_col_plan(true);
    REQUIRE(run(p) == multi_col_truth());
}

RJ_CASE(build_right, "Build on right") {
    Plan p = multi_col_plan(false);
    REQUIRE(run(p) == multi_col_truth());
}

// beyond the reference's cases: many pages per column through the shim (page pointer
// marshalling, staging chunks) and the std::runtime_error contract
RJ_CASE(many_pages, "100k-row join through the shim") {
    std::vector<Row> a, b;
    for (int i = 0; i < 100000; ++i) a.push_back({int32_t(i * 7 % 100003)});
    for (int i = 0; i < 50000; ++i) b.push_back({int32_t(i * 2)});
    Plan p = simple_plan(a, b);
    auto rows = run(p);
    size_t expect = 0;
    std::vector<char> seen(200000, 0);
    for (auto& r : b) seen[std::get<int32_t>(r[0])] = 1;
    for (auto& r : a) {
        int32_t k = std::get<int32_t>(r[0]);
        if (k < 200000 && seen[k]) ++expect;
    }
    REQUIRE(rows.size() == expect);
    for (auto& r : rows) REQUIRE(r[0] == r[1]);
}

RJ_CASE(error_contract, "errors are std::runtime_error") {
    Plan p = simple_plan({{1}}, {{1}});
    p.root = 17;
    void* ctx = Contest::build_context();
    bool  threw = false;
    try {
        (void)Contest::execute(p, ctx);
    } catch (const std::runtime_error&) {
        threw = true;
    }
    Contest::destroy_context(ctx);
    REQUIRE(threw);
}

// ---------------------------------------------------------------- large case
// Not one of the reference's unit cases: 2 M x 3 M PK-FK through the same shim, so that the
// page marshalling (thousands of individually allocated Pages in, `new Page`s out), the
// multi-pass partitioner and the VARCHAR materialisation are exercised from C++ as well.
// R(key INT32 unique, pay INT32 = 7*key+1, name VARCHAR = "s<key>"), S(key INT32 = FK, pay INT32 = row).
static Column int32_column(const std::vector<int32_t>& v) {
    Column col(DataType::INT32);
    const size_t cap = 1984;  // reference include/plan.h:205 for a NULL-free INT32 page
    for (size_t b = 0; b < v.size(); b += cap) {
        size_t     n = std::min(cap, v.size() - b);
        std::byte* p = col.new_page()->data;
        memset(p, 0, PAGE_SIZE);
        put16(p, uint16_t(n));
        put16(p + 2, uint16_t(n));
        memcpy(p + 4, v.data() + b, n * 4);
        size_t nb = (n + 7) / 8;
        memset(p + PAGE_SIZE - nb, 0xff, nb);
        if (n % 8) *reinterpret_cast<uint8_t*>(p + PAGE_SIZE - 1) = uint8_t((1u << (n % 8)) - 1u);
    }
    return col;
}

RJ_CASE(large_join, "Large PK-FK join with VARCHAR payload") {
    const uint32_t nR = 2'000'000, nS = 3'000'000;
    std::vector<int32_t> rk(nR), rp(nR), sk(nS), sp(nS);
    for (uint32_t i = 0; i < nR; ++i) {
        rk[i] = int32_t((uint64_t(i) * 2654435761ull + 17) % nR);  // a bijection of [0, nR)
        rp[i] = rk[i] * 7 + 1;
    }
    for (uint32_t j = 0; j < nS; ++j) {
        sk[j] = int32_t((uint64_t(j) * 40503ull + 5) % nR);
        sp[j] = int32_t(j);
    }
    std::vector<Row> names(nR, Row(1));
    for (uint32_t i = 0; i < nR; ++i) names[i][0] = "s" + std::to_string(rk[i]);
    ColumnarTable R, S;
    R.num_rows = nR;
    R.columns.push_back(int32_column(rk));
    R.columns.push_back(int32_column(rp));
    {
        ColumnarTable nm = to_columnar(names, {DataType::VARCHAR});
        R.columns.push_back(std::move(nm.columns[0]));
    }
    S.num_rows = nS;
    S.columns.push_back(int32_column(sk));
    S.columns.push_back(int32_column(sp));

    Plan p;
    p.inputs.push_back(std::move(R));
    p.inputs.push_back(std::move(S));
    size_t r = p.new_scan_node(0, {{0, DataType::INT32}, {1, DataType::INT32}, {2, DataType::VARCHAR}});
    size_t s = p.new_scan_node(1, {{0, DataType::INT32}, {1, DataType::INT32}});
    // output: S.pay, R.name, R.key, R.pay
    p.root = p.new_join_node(true, r, s, 0, 0,
                             {{4, DataType::INT32}, {2, DataType::VARCHAR}, {0, DataType::INT32}, {1, DataType::INT32}});
    void*         ctx = Contest::build_context();
    ColumnarTable out = Contest::execute(p, ctx);
    Contest::destroy_context(ctx);
    REQUIRE(out.num_rows == nS);
    REQUIRE(out.columns.size() == 4);
    std::vector<Row> rows = from_columnar(out);
    std::vector<uint8_t> seen(nS, 0);
    for (const Row& row : rows) {
        int32_t            j = std::get<int32_t>(row[0]);
        const std::string& nm = std::get<std::string>(row[1]);
        int32_t            k = std::get<int32_t>(row[2]);
        int32_t            pay = std::get<int32_t>(row[3]);
        REQUIRE(j >= 0 && uint32_t(j) < nS && !seen[j]);
        seen[j] = 1;
        REQUIRE(k == sk[j]);
        REQUIRE(pay == k * 7 + 1);
        REQUIRE(nm == "s" + std::to_string(k));
    }
}

int main() {
    int failed = 0;
    for (auto& c : cases()) {
        try {
            c.fn();
            printf("ok      %s\n", c.name);
        } catch (const std::exception& e) {
            ++failed;
            printf("FAILED  %s: %s\n", c.name, e.what());
        }
    }
    printf("%zu cases, %d failed\n", cases().size(), failed);
    return failed ? 1 : 0;
}
