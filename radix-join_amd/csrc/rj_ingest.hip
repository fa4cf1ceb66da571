// rj_ingest.hip — Table::from_csv on the device: CSV text -> typed columns ->
// filter -> Page images in HBM, i.e. the resident table a ScanNode reads without any upload.
//
// Replaces, for the harness's CSV dialect (escape '\\', separator ',', no header, no trailing
// comma: reference src/build_table.cpp:231):
//   CSVParser::execute / finish     src/csv_parser.cpp:3-175      k_csv_trans .. k_csv_emit
//   TableParser::on_field           src/build_table.cpp:31-76     k_csv_ints, k_csv_f64, k_csv_strlen
//   Comparison / LogicalOperation   src/statement.cpp:8-135,186-201 (over include/inner_column.h:
//                                   170-324, :372-562 for strings; LIKE: statement.h:118-161) k_ing_filter
//   from_inner_to_column +          src/build_table.cpp:94-119,
//   ColumnInserter<T>, <string>     include/plan.h:151-335        k_ing_next_*, k_hop_*, k_ing_walk_all, k_ing_expand,
//                                                                 k_ing_pages_fixed / _varchar
//
// The parser is a three-state machine — unquoted / quoted / quoted with an active backslash
// pending — whose only cross-record dependency is the quote state.  It is made parallel the
// classic way: every thread walks a 256-byte segment from each of the three possible start
// states (k_csv_trans), the 6-bit transitions are composed (per 64 KiB super-segment, then one
// thread over the super-segments), and with the true start states known a second walk counts,
// and a third one records, the unquoted separators: fend[row * n_cols + col] = position of the
// separator that ends the field.  A field always starts unquoted, so every later step (typed
// parse, string length, string copy) decodes its field on its own.
//
// The page-fill rules are sequential in the reference; here the first row of the NEXT page is
// computed for every row independently — "row j does not fit a page that starts at row i" is a
// monotone predicate over prefix sums (values, characters), found by binary search — and the chain
// of page starts is then shortened by pointer jumping, walked 32 hops at a time by one thread per
// column and filled in by one thread per 32-hop stretch (k_hop_*, k_ing_walk_all, k_ing_expand).  The
// pages written are the ones ColumnInserter produces, byte for byte where the reference defines
// the bytes.
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cstring>
#include <string>

#include "rj_fp64.hpp"
#include "rj_internal.hpp"

namespace rj {

namespace {

// 128-bit powers of five for the FP64 parser (rj_fp64.hpp), on the device and on the host
__device__ const uint64_t d_pow5[] = {
#include "rj_pow5_table.inc"
};
const uint64_t h_pow5[] = {
#include "rj_pow5_table.inc"
};

constexpr uint32_t SEG = 256;   // bytes a thread walks in the structural passes
constexpr uint32_t SUP = 256;   // segments per super-segment
constexpr uint32_t NULL_LEN = 0xffffffffu;
constexpr uint32_t VC_INLINE_MAX = PAGE_BYTES - 7;  // longest string of a normal page (plan.h:303)
constexpr uint32_t VC_PIECE = PAGE_BYTES - 4;       // characters per long-string page (plan.h:267)

#define RJ_ILAUNCH(L, NAME, KERNEL, GRID, BLOCK, ...)                                          \
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

// ------------------------------------------------------------------ the parser's state machine
// 0 unquoted, 1 quoted, 2 quoted + the previous byte was an ACTIVE backslash.  In quotes a
// backslash followed by '"' or '\\' stands for that character, otherwise for itself — either way
// the byte behind it does not change the quote state (csv_parser.cpp:144-160); a '"' toggles
// (:126-143, escape_ != '"'); outside quotes a backslash is an ordinary character.
__device__ __forceinline__ uint32_t csv_next(uint32_t st, uint8_t c) {
    if (st == 0) return c == '"' ? 1u : 0u;
    if (st == 1) return c == '"' ? 0u : (c == '\\' ? 2u : 1u);
    return 1u;
}

// the bytes [b, e) of the text, 16 at a time: b is a multiple of the segment size and the text starts
// on an allocation boundary, so the vector loads are aligned (one load instruction per 16 bytes —
// byte loads, every lane in a line of its own, kept the memory pipeline busy 16 times as long)
template <class F>
__device__ __forceinline__ void for_each_byte(const uint8_t* t, uint32_t b, uint32_t e, F&& f) {
    uint32_t i = b;
    for (; i + 16 <= e; i += 16) {
        const uint4    v = *reinterpret_cast<const uint4*>(t + i);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) f(i + k, (uint8_t)(w[k >> 2] >> (8 * (k & 3))));
    }
    for (; i < e; ++i) f(i, t[i]);
}

// transition of a run of bytes: to[s] packed two bits each
__device__ __forceinline__ uint32_t tr_apply(uint32_t tr, uint32_t s) { return (tr >> (2 * s)) & 3u; }
__device__ __forceinline__ uint32_t tr_compose(uint32_t first, uint32_t then) {  // then(first(s))
    return tr_apply(then, tr_apply(first, 0)) | (tr_apply(then, tr_apply(first, 1)) << 2) | (tr_apply(then, tr_apply(first, 2)) << 4);
}

__global__ __launch_bounds__(256) void k_csv_trans(const uint8_t* t, uint32_t n, uint32_t n_seg, uint8_t* trans) {
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= n_seg) return;
    const uint32_t b = seg * SEG, e = min(n, b + SEG);
    uint32_t       s0 = 0, s1 = 1, s2 = 2;
    for_each_byte(t, b, e, [&](uint32_t, uint8_t c) {
        s0 = csv_next(s0, c);
        s1 = csv_next(s1, c);
        s2 = csv_next(s2, c);
    });
    trans[seg] = (uint8_t)(s0 | (s1 << 2) | (s2 << 4));
}

// per super-segment: compose its segments' transitions (WRITE = false), or — the super-segment's
// start state known — hand every segment its start state (WRITE = true)
template <bool WRITE>
__global__ __launch_bounds__(256) void k_csv_sup(const uint8_t* trans, uint32_t n_seg, uint32_t n_sup, uint8_t* sup_trans,
                                                 const uint8_t* sup_start, uint8_t* seg_start) {
    const uint32_t sp = blockIdx.x * blockDim.x + threadIdx.x;
    if (sp >= n_sup) return;
    const uint32_t b = sp * SUP, e = min(n_seg, b + SUP);
    if (!WRITE) {
        uint32_t tr = 0u | (1u << 2) | (2u << 4);  // identity
        for (uint32_t s = b; s < e; ++s) tr = tr_compose(tr, trans[s]);
        sup_trans[sp] = (uint8_t)tr;
    } else {
        uint32_t st = sup_start[sp];
        for (uint32_t s = b; s < e; ++s) {
            seg_start[s] = (uint8_t)st;
            st = tr_apply(trans[s], st);
        }
    }
}

// one thread: start state of every super-segment; info[0] = the state the text ends in
__global__ void k_csv_sup_scan(const uint8_t* sup_trans, uint32_t n_sup, uint8_t* sup_start, uint32_t* info) {
    if (blockIdx.x || threadIdx.x) return;
    uint32_t st = 0;
    for (uint32_t s = 0; s < n_sup; ++s) {
        sup_start[s] = (uint8_t)st;
        st = tr_apply(sup_trans[s], st);
    }
    info[0] = st;
}

// Walk one segment with its true start state and report the unquoted separators: on_comma(pos),
// on_record_end(pos).  "\r\n" is ONE record end, at the '\r' (csv_parser.cpp:81-88): an unquoted
// '\n' right behind a '\r' is skipped (the '\r' left the state unquoted, so it was one too).
template <class FC, class FR>
__device__ __forceinline__ void csv_walk(const uint8_t* t, uint32_t n, uint32_t seg, uint32_t st, FC&& on_comma, FR&& on_rec) {
    const uint32_t b = seg * SEG, e = min(n, b + SEG);
    uint8_t        prev = b ? t[b - 1] : (uint8_t)0;
    for_each_byte(t, b, e, [&](uint32_t i, uint8_t c) {
        if (st == 0) {
            if (c == ',')
                on_comma(i);
            else if (c == '\r' || (c == '\n' && prev != '\r'))
                on_rec(i);
        }
        st = csv_next(st, c);
        prev = c;
    });
}

// per segment: record ends, and the commas behind the last of them (bit 31: the segment has one)
__global__ __launch_bounds__(256) void k_csv_count(const uint8_t* t, uint32_t n, uint32_t n_seg, const uint8_t* seg_start,
                                                   uint32_t* seg_rec, uint32_t* seg_tail) {
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= n_seg) return;
    uint32_t nrec = 0, tail = 0;
    csv_walk(t, n, seg, seg_start[seg], [&](uint32_t) { ++tail; }, [&](uint32_t) {
        ++nrec;
        tail = 0;
    });
    seg_rec[seg] = nrec;
    seg_tail[seg] = tail | (nrec ? 0x80000000u : 0u);
}

// the same two-level prefix for (records before the segment, column the segment starts in)
template <bool WRITE>
__global__ __launch_bounds__(256) void k_csv_sup_rows(uint32_t* seg_rec, uint32_t* seg_tail, uint32_t n_seg, uint32_t n_sup,
                                                      uint32_t* sup_rec, uint32_t* sup_tail, const uint32_t* sup_row0,
                                                      const uint32_t* sup_col0) {
    const uint32_t sp = blockIdx.x * blockDim.x + threadIdx.x;
    if (sp >= n_sup) return;
    const uint32_t b = sp * SUP, e = min(n_seg, b + SUP);
    if (!WRITE) {
        uint32_t rec = 0, tail = 0, has = 0;
        for (uint32_t s = b; s < e; ++s) {
            rec += seg_rec[s];
            const uint32_t tl = seg_tail[s];
            if (tl & 0x80000000u) {
                has = 0x80000000u;
                tail = tl & 0x7fffffffu;
            } else {
                tail += tl;
            }
        }
        sup_rec[sp] = rec;
        sup_tail[sp] = tail | has;
    } else {  // in place: seg_rec -> records before the segment, seg_tail -> its start column
        uint32_t row = sup_row0[sp], col = sup_col0[sp];
        for (uint32_t s = b; s < e; ++s) {
            const uint32_t rec = seg_rec[s], tl = seg_tail[s];
            seg_rec[s] = row;
            seg_tail[s] = col;
            row += rec;
            col = (tl & 0x80000000u) ? (tl & 0x7fffffffu) : col + tl;
        }
    }
}
__global__ void k_csv_sup_rows_scan(const uint32_t* sup_rec, const uint32_t* sup_tail, uint32_t n_sup, uint32_t* sup_row0,
                                    uint32_t* sup_col0, uint32_t* info) {
    if (blockIdx.x || threadIdx.x) return;
    uint32_t row = 0, col = 0;
    for (uint32_t s = 0; s < n_sup; ++s) {
        sup_row0[s] = row;
        sup_col0[s] = col;
        row += sup_rec[s];
        const uint32_t tl = sup_tail[s];
        col = (tl & 0x80000000u) ? (tl & 0x7fffffffu) : col + tl;
    }
    info[1] = row;  // records
    info[2] = col;  // fields of an unfinished last record (the host appended a '\n': must be 0)
}

// fend[row * n_cols + col] = position of the separator that ends the field; info[3] counts
// records with another number of fields than n_cols (CSVParser::InconsistentColumns)
__global__ __launch_bounds__(256) void k_csv_emit(const uint8_t* t, uint32_t n, uint32_t n_seg, const uint8_t* seg_start,
                                                  const uint32_t* seg_row, const uint32_t* seg_col, uint32_t n_cols,
                                                  uint32_t* fend, uint32_t* info) {
    const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= n_seg) return;
    uint32_t row = seg_row[seg], col = seg_col[seg], bad = 0;
    csv_walk(
        t, n, seg, seg_start[seg],
        [&](uint32_t pos) {
            if (col < n_cols) fend[(size_t)row * n_cols + col] = pos;
            ++col;
        },
        [&](uint32_t pos) {
            if (col < n_cols) fend[(size_t)row * n_cols + col] = pos;
            bad += col + 1 != n_cols;
            ++row;
            col = 0;
        });
    if (bad) atomicAdd(&info[3], bad);
}

// ------------------------------------------------------------------ one field at a time
struct Field {
    uint32_t beg, end;
};
__device__ __forceinline__ Field field_of(const uint8_t* t, const uint32_t* fend, uint32_t n_cols, uint32_t row, uint32_t col) {
    const size_t idx = (size_t)row * n_cols + col;
    Field        f{0u, fend[idx]};
    if (idx) {
        const uint32_t pe = fend[idx - 1];
        f.beg = pe + 1;
        if (t[pe] == '\r' && f.beg < f.end + 1 && t[f.beg] == '\n') ++f.beg;  // "\r\n"
    }
    return f;
}
// the characters TableParser::on_field sees (csv_parser.cpp:64-160): quotes dropped, escapes applied
struct FieldChars {
    const uint8_t* t;
    uint32_t       pos, end, st;
    __device__ __forceinline__ bool next(uint8_t& c) {
        while (pos < end) {
            const uint8_t x = t[pos++];
            if (st == 0) {
                if (x == '"') {
                    st = 1;
                    continue;
                }
                c = x;
                return true;
            }
            if (x == '"') {
                st = 0;
                continue;
            }
            if (x == '\\' && pos < end && (t[pos] == '"' || t[pos] == '\\')) {
                c = t[pos++];
                return true;
            }
            c = x;
            return true;
        }
        return false;
    }
};

// INT32 / INT64 columns: std::from_chars (build_table.cpp:37-56) — optional '-', at least one
// digit, stops at the first other character; an empty field is NULL (:34-35).  info[4] counts
// "parse integer error"s.
template <int W>
__global__ __launch_bounds__(256) void k_csv_ints(const uint8_t* t, const uint32_t* fend, uint32_t n_cols, uint32_t col,
                                                  uint32_t n_rows, uint8_t* values, uint8_t* valid, uint32_t* info) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const Field f = field_of(t, fend, n_cols, r, col);
    FieldChars  it{t, f.beg, f.end, 0};
    uint8_t     c;
    bool        any = it.next(c), neg = false, over = false;
    uint32_t    digits = 0;
    uint64_t    v = 0;
    const uint64_t lim = W == 4 ? 2147483648ull : 9223372036854775808ull;
    if (any && c == '-') {
        neg = true;
        if (!it.next(c)) c = 0;
    }
    if (any)
        while (c >= '0' && c <= '9') {
            const uint32_t d = c - '0';
            if (v > (lim - d) / 10) over = true;
            if (!over) v = v * 10 + d;
            ++digits;
            if (!it.next(c)) break;
        }
    const bool bad = any && (!digits || over || (!neg && v > lim - 1));
    if (bad) atomicAdd(&info[4], 1u);
    const int64_t sv = neg ? (int64_t)(0ull - v) : (int64_t)v;
    if (W == 4)
        reinterpret_cast<int32_t*>(values)[r] = any ? (int32_t)sv : 0;
    else
        reinterpret_cast<int64_t*>(values)[r] = any ? sv : 0;
    valid[r] = any ? 1 : 0;
}

// FP64 columns: std::from_chars(double) (build_table.cpp:57-64) — the nearest double of the
// decimal text (rj_fp64.hpp).  Fields outside the plain number grammar ("inf", "nan", a number
// followed by other characters ...) and the few whose rounding the 128-bit product cannot settle
// are listed in `hard` for the host's std::from_chars; info[5] counts "parse float error"s,
// info[6] the listed rows.
__global__ __launch_bounds__(256) void k_csv_f64(const uint8_t* t, const uint32_t* fend, uint32_t n_cols, uint32_t col,
                                                 uint32_t n_rows, uint64_t* values, uint8_t* valid, uint32_t* info, uint32_t* hard) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const Field f = field_of(t, fend, n_cols, r, col);
    FieldChars  probe{t, f.beg, f.end, 0};
    uint8_t     c;
    const bool  any = probe.next(c);
    uint64_t    bits = 0;
    if (any) {
        FieldChars it{t, f.beg, f.end, 0};
        const int  st = parse_fp64_field(it, d_pow5, &bits);
        if (st == FP64_RANGE) atomicAdd(&info[5], 1u);
        if (st == FP64_UNDECIDED) hard[atomicAdd(&info[6], 1u)] = r;
        if (st != FP64_PARSED) bits = 0;
    }
    values[r] = bits;
    valid[r] = any ? 1 : 0;
}
// where the listed rows' fields lie in the text (for the host), and their values coming back
__global__ __launch_bounds__(256) void k_csv_field_bounds(const uint8_t* t, const uint32_t* fend, uint32_t n_cols, uint32_t col,
                                                          const uint32_t* rows, uint32_t n, uint2* bounds) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Field f = field_of(t, fend, n_cols, rows[i], col);
    bounds[i] = make_uint2(f.beg, f.end);
}
__global__ __launch_bounds__(256) void k_csv_patch64(const uint32_t* rows, const uint64_t* vals, uint32_t n, uint64_t* values) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) values[rows[i]] = vals[i];
}

// VARCHAR columns: decoded length of every row's string, NULL_LEN for an empty field
__global__ __launch_bounds__(256) void k_csv_strlen(const uint8_t* t, const uint32_t* fend, uint32_t n_cols, uint32_t col,
                                                    uint32_t n_rows, uint32_t* len) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const Field f = field_of(t, fend, n_cols, r, col);
    FieldChars  it{t, f.beg, f.end, 0};
    uint8_t     c;
    uint32_t    n = 0;
    while (it.next(c)) ++n;
    len[r] = n ? n : NULL_LEN;
}

// ------------------------------------------------------------------ LIKE
// Comparison::like_match (reference include/statement.h:118-161): '%' -> ".*", '_' -> ".", every
// other character itself, and RE2 must match the WHOLE string with its defaults — UTF-8 (a '.' is
// one well-formed sequence) and no '.' for a newline.  A position-set automaton over the pattern's
// characters (at most 63): bit i = "the first i pattern characters are matched".
constexpr uint32_t LIKE_ANY = 0x80000000u, LIKE_RUN = 0x80000001u;
// one well-formed UTF-8 sequence off the field's characters, packed little-endian; false = none
__device__ __forceinline__ bool like_unit(FieldChars& it, uint32_t& cp, bool& bad) {
    uint8_t b0;
    if (!it.next(b0)) return false;
    uint32_t L = 0, lo = 0x80, hi = 0xBF;
    if (b0 < 0x80) L = 1;
    else if (b0 >= 0xC2 && b0 <= 0xDF) L = 2;
    else if (b0 == 0xE0) { L = 3; lo = 0xA0; }
    else if (b0 >= 0xE1 && b0 <= 0xEF) L = 3;
    else if (b0 == 0xF0) { L = 4; lo = 0x90; }
    else if (b0 >= 0xF1 && b0 <= 0xF3) L = 4;
    else if (b0 == 0xF4) { L = 4; hi = 0x8F; }
    if (!L) {
        bad = true;
        return true;
    }
    cp = b0;
    for (uint32_t k = 1; k < L; ++k) {
        uint8_t b;
        if (!it.next(b) || b < (k == 1 ? lo : 0x80u) || b > (k == 1 ? hi : 0xBFu)) {
            bad = true;
            return true;
        }
        cp |= (uint32_t)b << (8 * k);
    }
    return true;
}
__device__ __forceinline__ bool like_match(const uint32_t* tok, int m, FieldChars it) {
    if (m < 0) return false;  // the pattern is not UTF-8: RE2 would not compile it (:151-153)
    uint64_t st = 1;
    auto     closure = [&] {
        for (int i = 0; i < m; ++i)
            if (((st >> i) & 1u) && tok[i] == LIKE_RUN) st |= 1ull << (i + 1);
    };
    closure();
    uint32_t cp;
    bool     bad = false;
    while (like_unit(it, cp, bad)) {
        if (bad) return false;  // neither '.' nor a literal takes ill-formed bytes
        uint64_t nx = 0;
        for (int k = 0; k < m; ++k) {
            if (!((st >> k) & 1u)) continue;
            const uint32_t t = tok[k];
            if (t == LIKE_RUN)
                nx |= (uint64_t)(cp != '\n') << k;
            else if (t == LIKE_ANY)
                nx |= (uint64_t)(cp != '\n') << (k + 1);
            else
                nx |= (uint64_t)(t == cp) << (k + 1);
        }
        st = nx;
        closure();
        if (!st) return false;
    }
    return (st >> m) & 1u;
}

// ------------------------------------------------------------------ filter
struct DevFilterOp {
    int32_t        op, column;
    int64_t        ivalue;
    const uint8_t* bytes;  // device copy of the host bitmap / the string literal / the LIKE tokens (u32 each)
};
struct DevCol {
    const uint8_t*  values;  // INT32 / INT64 / FP64
    const uint8_t*  valid;   // fixed-width
    const uint32_t* len;     // VARCHAR (NULL_LEN = NULL)
    int32_t         type, pad;
};
constexpr int MAX_FILTER_OPS = 64, MAX_ING_COLS = 64;
struct FilterProg {
    DevFilterOp     ops[MAX_FILTER_OPS];
    DevCol          cols[MAX_ING_COLS];
    uint32_t        n_ops, n_cols;
    const uint8_t*  text;  // string comparisons decode the field itself
    const uint32_t* fend;
};
// one row of the bitmap arithmetic of statement.cpp:8-135,186-201: comparisons are false on NULL
// (inner_column.h:247-253: bitmap & cmp), NOT flips every bit, NULL rows included
__global__ __launch_bounds__(256) void k_ing_filter(const FilterProg* pp, uint32_t n_rows, uint32_t* sel) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const FilterProg& p = *pp;
    uint64_t          st = 0;  // a stack of bits
    uint32_t          sp = 0;
    for (uint32_t k = 0; k < p.n_ops; ++k) {
        const DevFilterOp& o = p.ops[k];
        if (o.op == RJ_F_AND || o.op == RJ_F_OR) {
            const uint64_t b = (st >> (sp - 1)) & 1u, a = (st >> (sp - 2)) & 1u;
            sp -= 2;
            st &= (1ull << sp) - 1ull;
            st |= (o.op == RJ_F_AND ? (a & b) : (a | b)) << sp;
            ++sp;
        } else if (o.op == RJ_F_NOT) {
            st ^= 1ull << (sp - 1);
        } else {
            uint64_t v = 0;
            if (o.op == RJ_F_HOST_BITMAP) {
                v = (o.bytes[r >> 3] >> (r & 7u)) & 1u;
            } else if (o.op == RJ_F_LIKE || o.op == RJ_F_NOT_LIKE) {  // false on NULL, both (inner_column.h:518-562)
                if (p.cols[o.column].len[r] != NULL_LEN) {
                    const Field f = field_of(p.text, p.fend, p.n_cols, r, (uint32_t)o.column);
                    const bool  hit = like_match(reinterpret_cast<const uint32_t*>(o.bytes), (int)o.ivalue, FieldChars{p.text, f.beg, f.end, 0});
                    v = o.op == RJ_F_LIKE ? hit : !hit;
                }
            } else if (o.op <= RJ_F_GEQ && p.cols[o.column].type == RJ_VARCHAR) {
                // std::string comparison (statement.cpp:117-126): unsigned bytes, then length
                const DevCol& c = p.cols[o.column];
                if (c.len[r] != NULL_LEN) {
                    const Field f = field_of(p.text, p.fend, p.n_cols, r, (uint32_t)o.column);
                    FieldChars  it{p.text, f.beg, f.end, 0};
                    const uint32_t lb = (uint32_t)o.ivalue;
                    int            d = 0;
                    uint8_t        ch;
                    uint32_t       k = 0;
                    for (;; ++k) {
                        const bool more = it.next(ch);
                        if (!more || k >= lb) {
                            d = more ? 1 : (k < lb ? -1 : 0);
                            break;
                        }
                        if (ch != o.bytes[k]) {
                            d = ch < o.bytes[k] ? -1 : 1;
                            break;
                        }
                    }
                    bool cmp;
                    switch (o.op) {
                    case RJ_F_EQ: cmp = d == 0; break;
                    case RJ_F_NEQ: cmp = d != 0; break;
                    case RJ_F_LT: cmp = d < 0; break;
                    case RJ_F_GT: cmp = d > 0; break;
                    case RJ_F_LEQ: cmp = d <= 0; break;
                    default: cmp = d >= 0; break;
                    }
                    v = cmp;
                }
            } else {
                const DevCol& c = p.cols[o.column];
                const bool    nn = c.type == RJ_VARCHAR ? c.len[r] != NULL_LEN : c.valid[r] != 0;
                if (o.op == RJ_F_IS_NULL)
                    v = !nn;
                else if (o.op == RJ_F_IS_NOT_NULL)
                    v = nn;
                else if (c.type == RJ_FP64) {  // IEEE comparisons against a double literal (statement.cpp:91-107)
                    const double x = reinterpret_cast<const double*>(c.values)[r], y = __longlong_as_double(o.ivalue);
                    bool         cmp = false;
                    switch (o.op) {
                    case RJ_F_EQ: cmp = x == y; break;
                    case RJ_F_NEQ: cmp = x != y; break;
                    case RJ_F_LT: cmp = x < y; break;
                    case RJ_F_GT: cmp = x > y; break;
                    case RJ_F_LEQ: cmp = x <= y; break;
                    default: cmp = x >= y; break;
                    }
                    v = nn && cmp;
                } else {
                    const int64_t x = c.type == RJ_INT32 ? (int64_t)reinterpret_cast<const int32_t*>(c.values)[r]
                                                         : reinterpret_cast<const int64_t*>(c.values)[r];
                    const int64_t y = c.type == RJ_INT32 ? (int64_t)(int32_t)o.ivalue : o.ivalue;  // statement.cpp:55
                    bool          cmp = false;
                    switch (o.op) {
                    case RJ_F_EQ: cmp = x == y; break;
                    case RJ_F_NEQ: cmp = x != y; break;
                    case RJ_F_LT: cmp = x < y; break;
                    case RJ_F_GT: cmp = x > y; break;
                    case RJ_F_LEQ: cmp = x <= y; break;
                    default: cmp = x >= y; break;
                    }
                    v = nn && cmp;
                }
            }
            st |= v << sp;
            ++sp;
        }
    }
    sel[r] = p.n_ops ? (uint32_t)(st & 1u) : 1u;
}

// ------------------------------------------------------------------ exclusive scan of u32, any n
// (in place; block totals -> launch_scan_bins -> added back)
constexpr uint32_t SCAN_TILE = 4096;
__global__ __launch_bounds__(1024) void k_scan_local(uint32_t* a, uint32_t n, uint32_t* block_tot) {
    __shared__ uint32_t s_w[16];
    const uint32_t      base = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
    uint32_t            v[4], sum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = base + k < n ? a[base + k] : 0u;
        sum += v[k];
    }
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    uint32_t       incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if (lane >= (uint32_t)off) incl += t;
    }
    if (lane == 63) s_w[wid] = incl;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
    for (uint32_t k = 0; k < 16; ++k) {
        if (k < wid) wbase += s_w[k];
        tot += s_w[k];
    }
    uint32_t run = wbase + incl - sum;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) a[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
}
__global__ __launch_bounds__(1024) void k_scan_add(uint32_t* a, uint32_t n, const uint32_t* block_off) {
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * 4, add = block_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + k < n) a[base + k] += add;
}

// ------------------------------------------------------------------ compaction of the selected rows
__global__ __launch_bounds__(256) void k_ing_compact_fixed(const uint32_t* sel, const uint32_t* excl, uint32_t n_rows, int width,
                                                           const uint8_t* values, const uint8_t* valid, uint8_t* o_values,
                                                           uint32_t* o_valid) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows || !sel[r]) return;
    const uint32_t o = excl[r];
    if (width == 4)
        reinterpret_cast<uint32_t*>(o_values)[o] = reinterpret_cast<const uint32_t*>(values)[r];
    else
        reinterpret_cast<uint64_t*>(o_values)[o] = reinterpret_cast<const uint64_t*>(values)[r];
    o_valid[o] = valid[r];
}
// VARCHAR: source row, decoded length, and what the fill rule adds up: (1 if non-NULL, characters —
// a long string counts a whole page, so that no page "fits" it)
__global__ __launch_bounds__(256) void k_ing_compact_vc(const uint32_t* sel, const uint32_t* excl, uint32_t n_rows,
                                                        const uint32_t* len, uint32_t* o_row, uint32_t* o_len, uint32_t* o_valid,
                                                        uint32_t* o_chars) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows || !sel[r]) return;
    const uint32_t o = excl[r], l = len[r];
    o_row[o] = r;
    o_len[o] = l;
    o_valid[o] = l != NULL_LEN;
    o_chars[o] = l == NULL_LEN ? 0u : (l > VC_INLINE_MAX ? PAGE_BYTES : l);
}

// ------------------------------------------------------------------ page boundaries
// Fixed-width (ColumnInserter<T>, plan.h:204-222), a page that starts at row i with data_end = DB:
//   row j is refused  <=>  valid:  B(j) + 4 > PAGE        (the literal 4 of :205, whatever sizeof(T))
//                          NULL:   B(j)     > PAGE,   B(j) = DB + W * values[i, j) + (j - i) / 8 + 1.
// B is monotone, so the first j with B(j) + 4 > PAGE is found by binary search; from there at most
// a handful of NULL rows may still slip in (a valid one is refused at once).
// vx = exclusive prefix of `valid` over the selected rows, vx[n] = total.
__global__ __launch_bounds__(256) void k_ing_next_fixed(const uint32_t* vx, const uint32_t* valid, uint32_t n, uint32_t W,
                                                        uint32_t* nxt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t DB = W, v0 = vx[i];
    auto           B = [&](uint32_t j) { return DB + W * (vx[j] - v0) + (j - i) / 8u + 1u; };
    uint32_t       lo = i, hi = n;  // first j in [i, n) with B(j) + 4 > PAGE, or n
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (B(mid) + 4u > PAGE_BYTES)
            hi = mid;
        else
            lo = mid + 1;
    }
    uint32_t j = lo;
    while (j < n && !valid[j] && B(j) <= PAGE_BYTES) ++j;
    nxt[i] = j;
}
// VARCHAR (ColumnInserter<std::string>, plan.h:302-331): row j is refused
//   <=>  4 + 2 * values[i, j] + chars[i, j] + (j - i) / 8 + 1 > PAGE     (both kinds: a NULL row adds
// nothing to the two sums), with the sums INCLUDING row j — monotone in j; a long string carries
// a page's worth of characters and is therefore always refused (it gets pages of its own).
__global__ __launch_bounds__(256) void k_ing_next_vc(const uint32_t* vx, const uint32_t* cx, uint32_t n, uint32_t* nxt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v0 = vx[i], c0 = cx[i];
    uint32_t       lo = i, hi = n;  // first refused j in [i, n), or n
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        const uint64_t g = 4ull + 2ull * (vx[mid + 1] - v0) + (uint64_t)(cx[mid + 1] - c0) + (mid - i) / 8u + 1u;
        if (g > PAGE_BYTES)
            hi = mid;
        else
            lo = mid + 1;
    }
    nxt[i] = lo;
}

struct IngPage {
    uint32_t first, nr, piece;  // piece: 0 = normal page, 1 + k = piece k of the long string in row `first`
};
// From page start to page start.  The chain 0 -> next(0) -> next(next(0)) ... is sequential (a hop is a
// dependent load), so it is shortened first: five rounds of pointer jumping (hop <- hop o hop, weights
// added; `next` is monotone, so the gathers of a round read almost in order) leave every row with its
// 32nd successor and the pages in between.  One thread per column then walks the 32-hop chain — all
// columns in one launch — and notes where every super-hop starts and how many pages lie before it;
// the hops inside each super-hop are re-walked by one thread per super-hop, in parallel, writing the
// page list.  A long string (VARCHAR) is a hop to the next row that weighs its number of pieces.
constexpr int HOP_ROUNDS = 5;
__device__ __forceinline__ uint2 hop0(const uint32_t* nxt, const uint32_t* len, uint32_t i) {
    if (len) {
        const uint32_t l = len[i];
        if (l != NULL_LEN && l > VC_INLINE_MAX) return make_uint2(i + 1u, (l + VC_PIECE - 1) / VC_PIECE);
    }
    return make_uint2(nxt[i], 1u);
}
__global__ __launch_bounds__(256) void k_hop_init(const uint32_t* nxt, const uint32_t* len, uint32_t n, uint2* hw) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    hw[i] = i < n ? hop0(nxt, len, i) : make_uint2(n, 0u);
}
__global__ __launch_bounds__(256) void k_hop_double(const uint2* in, uint32_t n, uint2* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    const uint2 a = in[i], b = in[a.x];
    out[i] = make_uint2(b.x, a.y + b.y);
}
struct WalkJob {
    const uint2* hw;       // {32nd successor, pages up to it} of every row
    uint2*       super;    // out: {row, pages before it} of every super-hop start
    uint32_t*    counts;   // out: [0] = super-hops, [1] = pages
    uint32_t     n, pad;
};
struct WalkJobs {
    WalkJob j[MAX_ING_COLS];
};
__global__ void k_ing_walk_all(WalkJobs jobs) {  // workgroup c = column c
    if (threadIdx.x) return;
    const WalkJob& w = jobs.j[blockIdx.x];
    uint32_t       ns = 0, pages = 0;
    for (uint32_t i = 0; i < w.n;) {
        w.super[ns++] = make_uint2(i, pages);
        const uint2 a = w.hw[i];
        pages += a.y;
        i = a.x;
    }
    w.counts[0] = ns;
    w.counts[1] = pages;
}
__global__ __launch_bounds__(256) void k_ing_expand(const uint32_t* nxt, const uint32_t* len, uint32_t n, const uint2* super,
                                                    uint32_t ns, IngPage* out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ns) return;
    const uint2    s = super[j];
    const uint32_t end = j + 1 < ns ? super[j + 1].x : n;
    uint32_t       np = s.y;
    for (uint32_t i = s.x; i < end;) {
        const uint2 h = hop0(nxt, len, i);
        if (h.x == i + 1u && len && len[i] != NULL_LEN && len[i] > VC_INLINE_MAX) {
            for (uint32_t p = 0; p < h.y; ++p) out[np + p] = IngPage{i, 0u, 1u + p};
        } else {
            out[np] = IngPage{i, h.x - i, 0u};
        }
        np += h.y;
        i = h.x;
    }
}

// ------------------------------------------------------------------ page writers
// one workgroup per page: values of the non-NULL rows densely from data_begin, the validity bitmap
// in the last (nr + 7) / 8 bytes, header {rows, values} (plan.h:179-190)
template <int W>
__global__ __launch_bounds__(256) void k_ing_pages_fixed(const IngPage* plist, const uint8_t* values, const uint32_t* valid,
                                                         uint8_t* pages) {
    __shared__ uint32_t s_w[4];
    const IngPage       pg = plist[blockIdx.x];
    uint8_t*            page = pages + (size_t)blockIdx.x * PAGE_BYTES;
    const uint32_t      lane = threadIdx.x & 63u, wid = threadIdx.x >> 6, nb = (pg.nr + 7) / 8;
    for (uint32_t k = threadIdx.x; k < PAGE_BYTES / 4; k += 256) reinterpret_cast<uint32_t*>(page)[k] = 0u;
    __syncthreads();
    uint8_t* bm = page + PAGE_BYTES - nb;
    uint32_t running = 0;
    for (uint32_t base = 0; base < pg.nr; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const bool     bit = i < pg.nr && valid[pg.first + i];
        const uint64_t mask = __ballot(bit);
        const uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (lane == 0) s_w[wid] = (uint32_t)__popcll(mask);
        __syncthreads();
        uint32_t wpre = 0, tot = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            if (k < wid) wpre += s_w[k];
            tot += s_w[k];
        }
        if (bit) {
            const uint32_t vi = running + wpre + pre;
            if (W == 4)
                *reinterpret_cast<uint32_t*>(page + 4 + (size_t)vi * 4) = reinterpret_cast<const uint32_t*>(values)[pg.first + i];
            else
                *reinterpret_cast<uint64_t*>(page + 8 + (size_t)vi * 8) = reinterpret_cast<const uint64_t*>(values)[pg.first + i];
        }
        running += tot;
        __syncthreads();
    }
    // The bitmap goes in AFTER the values, as save_page does (plan.h:186): for an 8-byte type the
    // inserter's test (`data_end + 4 + ...`, a literal 4, :205) lets the last value of a page reach
    // up to 4 bytes into the bitmap, and the bitmap then overwrites them — the reference's pages
    // hold that altered value, and so do these.
    for (uint32_t base = 0; base < pg.nr; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t mask = __ballot(i < pg.nr && valid[pg.first + i]);
        if (lane < 8) {
            const uint32_t byte_idx = (base >> 3) + wid * 8u + lane;
            if (byte_idx < nb) bm[byte_idx] = (uint8_t)(mask >> (lane * 8u));
        }
    }
    if (threadIdx.x == 0) {
        reinterpret_cast<uint16_t*>(page)[0] = (uint16_t)pg.nr;
        reinterpret_cast<uint16_t*>(page)[1] = (uint16_t)running;
    }
}

// VARCHAR page: {rows, values}, end offsets, characters (decoded from the CSV field by the thread
// that owns the row), bitmap; or one piece of a long string (plan.h:256-288)
__global__ __launch_bounds__(256) void k_ing_pages_vc(const IngPage* plist, const uint8_t* t, const uint32_t* fend, uint32_t n_cols,
                                                      uint32_t col, const uint32_t* o_row, const uint32_t* o_len, uint8_t* pages) {
    __shared__ uint32_t s_w[4], s_w2[4];
    const IngPage       pg = plist[blockIdx.x];
    uint8_t*            page = pages + (size_t)blockIdx.x * PAGE_BYTES;
    const uint32_t      lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    for (uint32_t k = threadIdx.x; k < PAGE_BYTES / 4; k += 256) reinterpret_cast<uint32_t*>(page)[k] = 0u;
    __syncthreads();
    if (pg.piece) {  // (one thread decodes the field up to this piece: long strings are rare)
        if (threadIdx.x == 0) {
            const uint32_t len = o_len[pg.first], skip = (pg.piece - 1) * VC_PIECE, nc = min(VC_PIECE, len - skip);
            const Field    f = field_of(t, fend, n_cols, o_row[pg.first], col);
            FieldChars     it{t, f.beg, f.end, 0};
            uint8_t        c;
            for (uint32_t k = 0; k < skip; ++k) (void)it.next(c);
            for (uint32_t k = 0; k < nc && it.next(c); ++k) page[4 + k] = c;
            reinterpret_cast<uint16_t*>(page)[0] = pg.piece == 1 ? (uint16_t)0xffff : (uint16_t)0xfffe;
            reinterpret_cast<uint16_t*>(page)[1] = (uint16_t)nc;
        }
        return;
    }
    // non-NULL rows of the page first: the characters start behind the offset array
    uint32_t cnt = 0;
    for (uint32_t i = threadIdx.x; i < pg.nr; i += 256) cnt += o_len[pg.first + i] != NULL_LEN;
    for (int off = 32; off; off >>= 1) cnt += __shfl_down(cnt, off);
    if (lane == 0) s_w[wid] = cnt;
    __syncthreads();
    const uint32_t nv = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    const uint32_t nb = (pg.nr + 7) / 8;
    uint8_t*       chars = page + 4 + (size_t)nv * 2;
    uint8_t*       bm = page + PAGE_BYTES - nb;
    uint32_t       run_v = 0, run_c = 0;
    for (uint32_t base = 0; base < pg.nr; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t l = i < pg.nr ? o_len[pg.first + i] : NULL_LEN;
        const bool     valid = l != NULL_LEN;
        const uint32_t len = valid ? l : 0u;
        const uint64_t mask = __ballot(valid);
        const uint32_t vpre = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        uint32_t       incl = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t x = __shfl_up(incl, off);
            if (lane >= (uint32_t)off) incl += x;
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
            *reinterpret_cast<uint16_t*>(page + 4 + (size_t)vi * 2) = (uint16_t)cend;
            const Field f = field_of(t, fend, n_cols, o_row[pg.first + i], col);
            FieldChars  it{t, f.beg, f.end, 0};
            uint8_t     c;
            uint8_t*    dst = chars + (cend - len);
            for (uint32_t k = 0; k < len && it.next(c); ++k) dst[k] = c;
        }
        if (lane < 8) {
            const uint32_t byte_idx = (base >> 3) + wid * 8u + lane;
            if (byte_idx < nb) bm[byte_idx] = (uint8_t)(mask >> (lane * 8u));
        }
        run_v += vtot;
        run_c += ctot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        reinterpret_cast<uint16_t*>(page)[0] = (uint16_t)pg.nr;
        reinterpret_cast<uint16_t*>(page)[1] = (uint16_t)nv;
    }
}

// in-place exclusive scan of a[0..n) on the device; a[n] (if has_total) receives the total
void device_scan(Context* ctx, const Launch& L, uint32_t* a, uint32_t n, uint32_t* total_at /* device, may be null */) {
    if (!n) {
        if (total_at) RJ_HIP(hipMemsetAsync(total_at, 0, 4, ctx->stream));
        return;
    }
    const uint32_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    BufP           tot = ctx->buf((size_t)nb * 4), off = ctx->buf(((size_t)nb + 1) * 4);
    RJ_ILAUNCH(L, "ingest_scan", k_scan_local, nb, 1024, a, n, tot->as<uint32_t>());
    launch_scan_bins(L, tot->as<uint32_t>(), nb, off->as<uint32_t>(), nullptr);
    RJ_ILAUNCH(L, "ingest_scan", k_scan_add, nb, 1024, a, n, off->as<uint32_t>());
    if (total_at) RJ_HIP(hipMemcpyAsync(total_at, off->as<uint32_t>() + nb, 4, hipMemcpyDeviceToDevice, ctx->stream));
}

uint32_t read_u32(Context* ctx, const uint32_t* dev) {
    uint32_t* h = static_cast<uint32_t*>(ctx->small_pinned());
    RJ_HIP(hipMemcpyAsync(h, dev, 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    return h[0];
}

}  // namespace

Table* table_from_csv(Context* ctx, const char* text, uint64_t n_bytes, uint64_t n_cols, const int32_t* col_type,
                      const rj_filter_op* filter, uint64_t n_filter_ops) {
    if (!n_cols || !col_type || (n_bytes && !text)) throw_fmt(RJ_ERR_ARG, "from_csv: bad arguments");
    if (n_cols > (uint64_t)MAX_ING_COLS) throw_fmt(RJ_ERR_UNSUPPORTED, "from_csv: more than %d columns", MAX_ING_COLS);
    if (n_filter_ops > (uint64_t)MAX_FILTER_OPS) throw_fmt(RJ_ERR_UNSUPPORTED, "from_csv: filter longer than %d operations", MAX_FILTER_OPS);
    if (n_bytes > 0xffffffe0ull) throw_fmt(RJ_ERR_UNSUPPORTED, "from_csv: more than 2^32 bytes of text");
    for (uint64_t c = 0; c < n_cols; ++c) {
        if (col_type[c] < RJ_INT32 || col_type[c] > RJ_VARCHAR) throw_fmt(RJ_ERR_ARG, "from_csv: bad column type");
    }
    {  // the filter must be a well-formed postfix program over columns it may touch
        int depth = 0;
        for (uint64_t k = 0; k < n_filter_ops; ++k) {
            const rj_filter_op& o = filter[k];
            if (o.op == RJ_F_AND || o.op == RJ_F_OR)
                depth -= 1;
            else if (o.op == RJ_F_NOT)
                depth -= 0;
            else if (o.op == RJ_F_LIKE || o.op == RJ_F_NOT_LIKE) {
                if (o.column < 0 || (uint64_t)o.column >= n_cols || col_type[o.column] != RJ_VARCHAR)
                    throw_fmt(RJ_ERR_ARG, "from_csv: LIKE wants a VARCHAR column");
                if (o.ivalue < 0 || o.ivalue > 4 * 63 || (o.ivalue && !o.bytes)) throw_fmt(RJ_ERR_UNSUPPORTED, "from_csv: LIKE pattern too long");
                depth += 1;
            } else if (o.op >= RJ_F_EQ && o.op <= RJ_F_HOST_BITMAP) {
                if (o.op == RJ_F_HOST_BITMAP) {
                    if (!o.bytes) throw_fmt(RJ_ERR_ARG, "from_csv: host bitmap leaf without a bitmap");
                } else {
                    if (o.column < 0 || (uint64_t)o.column >= n_cols) throw_fmt(RJ_ERR_ARG, "from_csv: filter column out of range");
                    if (o.op <= RJ_F_GEQ && col_type[o.column] == RJ_VARCHAR && (o.ivalue < 0 || o.ivalue > 0x7fffffff || (o.ivalue && !o.bytes)))
                        throw_fmt(RJ_ERR_ARG, "from_csv: string literal without bytes");
                }
                depth += 1;
            } else
                throw_fmt(RJ_ERR_ARG, "from_csv: bad filter opcode");
            if (depth < 1 || depth > 60) throw_fmt(RJ_ERR_ARG, "from_csv: malformed filter program");
        }
        if (n_filter_ops && depth != 1) throw_fmt(RJ_ERR_ARG, "from_csv: malformed filter program");
    }
    const Launch L = ctx->launch();
    // ---- the text in HBM; finish() ends an unterminated last record (csv_parser.cpp:164-175)
    const bool     add_nl = n_bytes && text[n_bytes - 1] != '\n' && text[n_bytes - 1] != '\r';
    const uint32_t n = (uint32_t)(n_bytes + (add_nl ? 1 : 0));
    std::unique_ptr<Table> tab(new rj_table());
    tab->ctx = ctx;
    tab->cols.resize(n_cols);
    for (uint64_t c = 0; c < n_cols; ++c) tab->cols[c].type = col_type[c];
    if (!n) return tab.release();
    BufP dtext = ctx->buf((size_t)n + 16);
    {
        constexpr size_t CH = (size_t)32 << 20;
        uint8_t*         stage = static_cast<uint8_t*>(ctx->staging(2 * CH));
        struct Ev {  // (destroyed on every way out)
            hipEvent_t e = nullptr;
            ~Ev() {
                if (e) (void)hipEventDestroy(e);
            }
        } ev[2];
        RJ_HIP(hipEventCreateWithFlags(&ev[0].e, hipEventDisableTiming));
        RJ_HIP(hipEventCreateWithFlags(&ev[1].e, hipEventDisableTiming));
        size_t k = 0;
        for (size_t o = 0; o < n_bytes; o += CH, ++k) {
            const size_t m = std::min(CH, (size_t)n_bytes - o);
            if (k >= 2) (void)hipEventSynchronize(ev[k & 1].e);
            {  // (the host workers share the copy: one thread moves ~10 GB/s, the link 50)
                uint8_t*    dst = stage + (k & 1) * CH;
                const char* srcp = text + o;
                parallel_for(m, (size_t)1 << 20, [&](size_t b0, size_t e0) { memcpy(dst + b0, srcp + b0, e0 - b0); });
            }
            RJ_HIP(hipMemcpyAsync(dtext->as<uint8_t>() + o, stage + (k & 1) * CH, m, hipMemcpyHostToDevice, ctx->stream));
            RJ_HIP(hipEventRecord(ev[k & 1].e, ctx->stream));
        }
        ctx->sync();
        if (add_nl) RJ_HIP(hipMemsetAsync(dtext->as<uint8_t>() + n_bytes, '\n', 1, ctx->stream));
    }
    const uint8_t* t = dtext->as<uint8_t>();
    // ---- structure: quote states, records, field ends
    const uint32_t n_seg = (n + SEG - 1) / SEG, n_sup = (n_seg + SUP - 1) / SUP;
    BufP           trans = ctx->buf(n_seg), seg_start = ctx->buf(n_seg), sup_trans = ctx->buf(n_sup), sup_start = ctx->buf(n_sup);
    BufP           info = ctx->buf(64);
    RJ_HIP(hipMemsetAsync(info->p, 0, 64, ctx->stream));
    uint32_t* dinfo = info->as<uint32_t>();
    RJ_ILAUNCH(L, "csv_structure", k_csv_trans, (n_seg + 255) / 256, 256, t, n, n_seg, trans->as<uint8_t>());
    RJ_ILAUNCH(L, "csv_structure", (k_csv_sup<false>), (n_sup + 255) / 256, 256, trans->as<uint8_t>(), n_seg, n_sup,
               sup_trans->as<uint8_t>(), (const uint8_t*)nullptr, (uint8_t*)nullptr);
    RJ_ILAUNCH(L, "csv_structure", k_csv_sup_scan, 1, 64, sup_trans->as<uint8_t>(), n_sup, sup_start->as<uint8_t>(), dinfo);
    RJ_ILAUNCH(L, "csv_structure", (k_csv_sup<true>), (n_sup + 255) / 256, 256, trans->as<uint8_t>(), n_seg, n_sup,
               (uint8_t*)nullptr, sup_start->as<uint8_t>(), seg_start->as<uint8_t>());
    BufP seg_rec = ctx->buf((size_t)n_seg * 4), seg_tail = ctx->buf((size_t)n_seg * 4);
    BufP sup_rec = ctx->buf((size_t)n_sup * 4), sup_tail = ctx->buf((size_t)n_sup * 4), sup_row0 = ctx->buf((size_t)n_sup * 4),
         sup_col0 = ctx->buf((size_t)n_sup * 4);
    RJ_ILAUNCH(L, "csv_structure", k_csv_count, (n_seg + 255) / 256, 256, t, n, n_seg, seg_start->as<uint8_t>(),
               seg_rec->as<uint32_t>(), seg_tail->as<uint32_t>());
    RJ_ILAUNCH(L, "csv_structure", (k_csv_sup_rows<false>), (n_sup + 255) / 256, 256, seg_rec->as<uint32_t>(),
               seg_tail->as<uint32_t>(), n_seg, n_sup, sup_rec->as<uint32_t>(), sup_tail->as<uint32_t>(), (const uint32_t*)nullptr,
               (const uint32_t*)nullptr);
    RJ_ILAUNCH(L, "csv_structure", k_csv_sup_rows_scan, 1, 64, sup_rec->as<uint32_t>(), sup_tail->as<uint32_t>(), n_sup,
               sup_row0->as<uint32_t>(), sup_col0->as<uint32_t>(), dinfo);
    RJ_ILAUNCH(L, "csv_structure", (k_csv_sup_rows<true>), (n_sup + 255) / 256, 256, seg_rec->as<uint32_t>(),
               seg_tail->as<uint32_t>(), n_seg, n_sup, (uint32_t*)nullptr, (uint32_t*)nullptr, sup_row0->as<uint32_t>(),
               sup_col0->as<uint32_t>());
    uint32_t* hinfo = static_cast<uint32_t*>(ctx->small_pinned());
    RJ_HIP(hipMemcpyAsync(hinfo, dinfo, 16, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    if (hinfo[0] != 0) throw_fmt(RJ_ERR_DATA, "CSV parse error");  // QuoteNotClosed
    const uint32_t n_rows = hinfo[1];
    if (hinfo[2] != 0) throw_fmt(RJ_ERR_DATA, "CSV parse error");
    if ((uint64_t)n_rows * n_cols > 0xfffffff0ull) throw_fmt(RJ_ERR_UNSUPPORTED, "from_csv: more than 2^32 fields");
    if (!n_rows) return tab.release();
    BufP fend = ctx->buf((size_t)n_rows * n_cols * 4);
    RJ_ILAUNCH(L, "csv_structure", k_csv_emit, (n_seg + 255) / 256, 256, t, n, n_seg, seg_start->as<uint8_t>(),
               seg_rec->as<uint32_t>(), seg_tail->as<uint32_t>(), (uint32_t)n_cols, fend->as<uint32_t>(), dinfo);
    if (read_u32(ctx, dinfo + 3) != 0) throw_fmt(RJ_ERR_DATA, "CSV parse error");  // InconsistentColumns
    // ---- typed columns of ALL rows (the reference's InnerTable, build_table.cpp:151-245)
    struct Typed {
        BufP values, valid, len;
    };
    std::vector<Typed> typed(n_cols);
    const uint32_t     rgrid = (n_rows + 255) / 256;
    std::unique_ptr<FilterProg> prog(new FilterProg());
    memset(prog.get(), 0, sizeof(FilterProg));
    for (uint64_t c = 0; c < n_cols; ++c) {
        Typed& ty = typed[c];
        if (col_type[c] == RJ_VARCHAR) {
            ty.len = ctx->buf((size_t)n_rows * 4);
            RJ_ILAUNCH(L, "csv_fields", k_csv_strlen, rgrid, 256, t, fend->as<uint32_t>(), (uint32_t)n_cols, (uint32_t)c, n_rows,
                       ty.len->as<uint32_t>());
        } else {
            const int W = col_type[c] == RJ_INT32 ? 4 : 8;
            ty.values = ctx->buf((size_t)n_rows * W);
            ty.valid = ctx->buf(n_rows);
            if (col_type[c] == RJ_FP64) {
                BufP hard = ctx->buf((size_t)n_rows * 4);
                RJ_HIP(hipMemsetAsync(dinfo + 6, 0, 4, ctx->stream));
                RJ_ILAUNCH(L, "csv_fields", k_csv_f64, rgrid, 256, t, fend->as<uint32_t>(), (uint32_t)n_cols, (uint32_t)c, n_rows,
                           ty.values->as<uint64_t>(), ty.valid->as<uint8_t>(), dinfo, hard->as<uint32_t>());
                const uint32_t n_hard = read_u32(ctx, dinfo + 6);
                if (n_hard) {
                    // the host's std::from_chars takes the fields the device left undecided
                    BufP bounds = ctx->buf((size_t)n_hard * 8), vals = ctx->buf((size_t)n_hard * 8);
                    RJ_ILAUNCH(L, "csv_fields", k_csv_field_bounds, (n_hard + 255) / 256, 256, t, fend->as<uint32_t>(), (uint32_t)n_cols,
                               (uint32_t)c, hard->as<uint32_t>(), n_hard, bounds->as<uint2>());
                    std::vector<uint2>    hb(n_hard);
                    std::vector<uint64_t> hv(n_hard, 0);
                    RJ_HIP(hipMemcpyAsync(hb.data(), bounds->p, (size_t)n_hard * 8, hipMemcpyDeviceToHost, ctx->stream));
                    ctx->sync();
                    std::atomic<uint32_t> bad{0};
                    parallel_for(n_hard, 1024, [&](size_t b0, size_t e0) {
                        std::string fld;
                        for (size_t i = b0; i < e0; ++i) {
                            // the characters on_field sees: quotes dropped, escapes applied (FieldChars)
                            fld.clear();
                            const uint32_t end = std::min<uint32_t>(hb[i].y, (uint32_t)n_bytes);
                            uint32_t       st = 0;
                            for (uint32_t p = hb[i].x; p < end;) {
                                const char x = text[p++];
                                if (st == 0) {
                                    if (x == '"') st = 1;
                                    else fld.push_back(x);
                                } else if (x == '"') {
                                    st = 0;
                                } else if (x == '\\' && p < end && (text[p] == '"' || text[p] == '\\')) {
                                    fld.push_back(text[p++]);
                                } else {
                                    fld.push_back(x);
                                }
                            }
                            double     v = 0;
                            const auto res = std::from_chars(fld.data(), fld.data() + fld.size(), v);
                            if (res.ec != std::errc()) ++bad;
                            memcpy(&hv[i], &v, 8);
                        }
                    });
                    if (bad.load()) throw_fmt(RJ_ERR_DATA, "parse float error");
                    RJ_HIP(hipMemcpyAsync(vals->p, hv.data(), (size_t)n_hard * 8, hipMemcpyHostToDevice, ctx->stream));
                    RJ_ILAUNCH(L, "csv_fields", k_csv_patch64, (n_hard + 255) / 256, 256, hard->as<uint32_t>(), vals->as<uint64_t>(), n_hard,
                               ty.values->as<uint64_t>());
                    ctx->sync();  // (hv is pageable and local)
                }
            } else if (W == 4)
                RJ_ILAUNCH(L, "csv_fields", (k_csv_ints<4>), rgrid, 256, t, fend->as<uint32_t>(), (uint32_t)n_cols, (uint32_t)c,
                           n_rows, ty.values->as<uint8_t>(), ty.valid->as<uint8_t>(), dinfo);
            else
                RJ_ILAUNCH(L, "csv_fields", (k_csv_ints<8>), rgrid, 256, t, fend->as<uint32_t>(), (uint32_t)n_cols, (uint32_t)c,
                           n_rows, ty.values->as<uint8_t>(), ty.valid->as<uint8_t>(), dinfo);
        }
        prog->cols[c] = DevCol{ty.values ? ty.values->as<uint8_t>() : nullptr, ty.valid ? ty.valid->as<uint8_t>() : nullptr,
                               ty.len ? ty.len->as<uint32_t>() : nullptr, col_type[c], 0};
    }
    if (read_u32(ctx, dinfo + 4) != 0) throw_fmt(RJ_ERR_DATA, "parse integer error");
    if (read_u32(ctx, dinfo + 5) != 0) throw_fmt(RJ_ERR_DATA, "parse float error");
    // ---- filter -> selection -> output row of every selected row
    std::vector<BufP> bitmaps;
    prog->n_ops = (uint32_t)n_filter_ops;
    prog->n_cols = (uint32_t)n_cols;
    prog->text = t;
    prog->fend = fend->as<uint32_t>();
    for (uint64_t k = 0; k < n_filter_ops; ++k) {
        prog->ops[k] = DevFilterOp{filter[k].op, filter[k].column, filter[k].ivalue, nullptr};
        size_t nb = 0;  // bytes that travel with the leaf: a host bitmap, or a string literal
        if (filter[k].op == RJ_F_LIKE || filter[k].op == RJ_F_NOT_LIKE) {
            // the pattern, one token per character ('%' / '_' / the character's UTF-8 bytes packed)
            std::vector<uint32_t> tok;
            const uint8_t*        pat = filter[k].bytes;
            const size_t          pn = (size_t)filter[k].ivalue;
            bool                  utf8 = true;
            for (size_t i = 0; i < pn && utf8;) {
                const uint8_t b0 = pat[i];
                size_t        len = 0;
                uint8_t       lo = 0x80, hi = 0xBF;
                if (b0 < 0x80) len = 1;
                else if (b0 >= 0xC2 && b0 <= 0xDF) len = 2;
                else if (b0 == 0xE0) { len = 3; lo = 0xA0; }
                else if (b0 >= 0xE1 && b0 <= 0xEF) len = 3;
                else if (b0 == 0xF0) { len = 4; lo = 0x90; }
                else if (b0 >= 0xF1 && b0 <= 0xF3) len = 4;
                else if (b0 == 0xF4) { len = 4; hi = 0x8F; }
                if (!len || i + len > pn) {
                    utf8 = false;
                    break;
                }
                uint32_t cp = b0;
                for (size_t q = 1; q < len; ++q) {
                    const uint8_t b = pat[i + q];
                    if (b < (q == 1 ? lo : 0x80) || b > (q == 1 ? hi : 0xBF)) utf8 = false;
                    cp |= (uint32_t)b << (8 * q);
                }
                tok.push_back(cp == '%' ? LIKE_RUN : (cp == '_' ? LIKE_ANY : cp));
                i += len;
            }
            if (utf8 && tok.size() > 63) throw_fmt(RJ_ERR_UNSUPPORTED, "from_csv: LIKE pattern of more than 63 characters");
            prog->ops[k].ivalue = utf8 ? (int64_t)tok.size() : -1;
            if (utf8 && !tok.empty()) {
                bitmaps.push_back(ctx->buf(tok.size() * 4));
                RJ_HIP(hipMemcpy(bitmaps.back()->p, tok.data(), tok.size() * 4, hipMemcpyHostToDevice));
                prog->ops[k].bytes = bitmaps.back()->as<uint8_t>();
            }
            continue;
        }
        if (filter[k].op == RJ_F_HOST_BITMAP)
            nb = ((size_t)n_rows + 7) / 8;
        else if (filter[k].op <= RJ_F_GEQ && col_type[filter[k].column] == RJ_VARCHAR)
            nb = (size_t)filter[k].ivalue;
        if (nb) {
            bitmaps.push_back(ctx->buf(nb));
            RJ_HIP(hipMemcpyAsync(bitmaps.back()->p, filter[k].bytes, nb, hipMemcpyHostToDevice, ctx->stream));
            prog->ops[k].bytes = bitmaps.back()->as<uint8_t>();
        }
    }
    BufP dprog = ctx->buf(sizeof(FilterProg));
    RJ_HIP(hipMemcpyAsync(dprog->p, prog.get(), sizeof(FilterProg), hipMemcpyHostToDevice, ctx->stream));
    ctx->sync();  // (the pageable sources above: host bitmaps, prog)
    BufP sel = ctx->buf((size_t)n_rows * 4), excl = ctx->buf(((size_t)n_rows + 1) * 4);
    RJ_ILAUNCH(L, "ingest_filter", k_ing_filter, rgrid, 256, dprog->as<FilterProg>(), n_rows, sel->as<uint32_t>());
    RJ_HIP(hipMemcpyAsync(excl->p, sel->p, (size_t)n_rows * 4, hipMemcpyDeviceToDevice, ctx->stream));
    device_scan(ctx, L, excl->as<uint32_t>(), n_rows, excl->as<uint32_t>() + n_rows);
    const uint32_t n_out = read_u32(ctx, excl->as<uint32_t>() + n_rows);
    tab->num_rows = n_out;
    if (!n_out) return tab.release();
    // ---- pages (from_inner_to_column, build_table.cpp:94-119): per column the compacted values, the
    // "first row of the next page" of every row, the page starts (pointer jumping, then ONE launch that
    // walks all columns' 32-hop chains, then one thread per super-hop), and a page writer per column
    const uint32_t ogrid = (n_out + 255) / 256;
    std::vector<const void*> dev_pages(n_cols, nullptr);
    std::vector<uint64_t>    n_pages(n_cols, 0);
    std::vector<BufP>        page_bufs(n_cols);
    struct ColWork {
        BufP vx, o_valid, nxt, o_values, o_row, o_len, cx, plist, hw_a, hw_b, super;
    };
    std::vector<ColWork> work(n_cols);
    BufP                 np_dev = ctx->buf((size_t)n_cols * 8);  // per column: super-hops, pages
    std::unique_ptr<WalkJobs> jobs(new WalkJobs());
    memset(jobs.get(), 0, sizeof(WalkJobs));
    for (uint64_t c = 0; c < n_cols; ++c) {
        ColWork& w = work[c];
        w.vx = ctx->buf(((size_t)n_out + 1) * 4);
        w.o_valid = ctx->buf((size_t)n_out * 4);
        w.nxt = ctx->buf((size_t)n_out * 4);
        if (col_type[c] != RJ_VARCHAR) {
            const int W = col_type[c] == RJ_INT32 ? 4 : 8;
            w.o_values = ctx->buf((size_t)n_out * W);
            RJ_ILAUNCH(L, "ingest_compact", k_ing_compact_fixed, rgrid, 256, sel->as<uint32_t>(), excl->as<uint32_t>(), n_rows, W,
                       typed[c].values->as<uint8_t>(), typed[c].valid->as<uint8_t>(), w.o_values->as<uint8_t>(),
                       w.o_valid->as<uint32_t>());
            RJ_HIP(hipMemcpyAsync(w.vx->p, w.o_valid->p, (size_t)n_out * 4, hipMemcpyDeviceToDevice, ctx->stream));
            device_scan(ctx, L, w.vx->as<uint32_t>(), n_out, w.vx->as<uint32_t>() + n_out);
            RJ_ILAUNCH(L, "ingest_next", k_ing_next_fixed, ogrid, 256, w.vx->as<uint32_t>(), w.o_valid->as<uint32_t>(), n_out,
                       (uint32_t)W, w.nxt->as<uint32_t>());
        } else {
            w.o_row = ctx->buf((size_t)n_out * 4);
            w.o_len = ctx->buf((size_t)n_out * 4);
            w.cx = ctx->buf(((size_t)n_out + 1) * 4);
            RJ_ILAUNCH(L, "ingest_compact", k_ing_compact_vc, rgrid, 256, sel->as<uint32_t>(), excl->as<uint32_t>(), n_rows,
                       typed[c].len->as<uint32_t>(), w.o_row->as<uint32_t>(), w.o_len->as<uint32_t>(), w.vx->as<uint32_t>(),
                       w.cx->as<uint32_t>());
            device_scan(ctx, L, w.vx->as<uint32_t>(), n_out, w.vx->as<uint32_t>() + n_out);
            device_scan(ctx, L, w.cx->as<uint32_t>(), n_out, w.cx->as<uint32_t>() + n_out);
            RJ_ILAUNCH(L, "ingest_next", k_ing_next_vc, ogrid, 256, w.vx->as<uint32_t>(), w.cx->as<uint32_t>(), n_out,
                       w.nxt->as<uint32_t>());
        }
        // the 32-hop chain of this column (pointer jumping, ping-pong between two arrays)
        w.hw_a = ctx->buf(((size_t)n_out + 1) * 8);
        w.hw_b = ctx->buf(((size_t)n_out + 1) * 8);
        const uint32_t* lenp = w.o_len ? w.o_len->as<uint32_t>() : nullptr;
        RJ_ILAUNCH(L, "ingest_walk", k_hop_init, (n_out + 256) / 256, 256, w.nxt->as<uint32_t>(), lenp, n_out, w.hw_a->as<uint2>());
        for (int r = 0; r < HOP_ROUNDS; ++r) {
            RJ_ILAUNCH(L, "ingest_walk", k_hop_double, (n_out + 256) / 256, 256, w.hw_a->as<uint2>(), n_out, w.hw_b->as<uint2>());
            std::swap(w.hw_a, w.hw_b);
        }
        w.super = ctx->buf(((size_t)n_out / (1u << HOP_ROUNDS) + 2) * 8);
        jobs->j[c] = WalkJob{w.hw_a->as<uint2>(), w.super->as<uint2>(), np_dev->as<uint32_t>() + 2 * c, n_out, 0u};
    }
    RJ_ILAUNCH(L, "ingest_walk", k_ing_walk_all, (uint32_t)n_cols, 64, *jobs);
    std::vector<uint32_t> np_host(2 * n_cols, 0);
    RJ_HIP(hipMemcpyAsync(np_host.data(), np_dev->p, (size_t)n_cols * 8, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
    for (uint64_t c = 0; c < n_cols; ++c) {
        ColWork&       w = work[c];
        const uint32_t ns = np_host[2 * c], np = np_host[2 * c + 1];
        w.plist = ctx->buf((size_t)std::max<uint32_t>(np, 1u) * sizeof(IngPage));
        if (ns)
            RJ_ILAUNCH(L, "ingest_walk", k_ing_expand, (ns + 255) / 256, 256, w.nxt->as<uint32_t>(),
                       w.o_len ? w.o_len->as<uint32_t>() : nullptr, n_out, w.super->as<uint2>(), ns, w.plist->as<IngPage>());
    }
    std::vector<BufP> vpages(n_cols);
    for (uint64_t c = 0; c < n_cols; ++c) {
        ColWork&       w = work[c];
        const uint32_t np = np_host[2 * c + 1];
        if (!np) continue;  // (cannot happen: n_out > 0 rows make at least one page)
        if (col_type[c] != RJ_VARCHAR) {
            page_bufs[c] = ctx->buf((size_t)np * PAGE_BYTES);
            if (col_type[c] == RJ_INT32)
                RJ_ILAUNCH(L, "ingest_pages", (k_ing_pages_fixed<4>), np, 256, w.plist->as<IngPage>(), w.o_values->as<uint8_t>(),
                           w.o_valid->as<uint32_t>(), page_bufs[c]->as<uint8_t>());
            else
                RJ_ILAUNCH(L, "ingest_pages", (k_ing_pages_fixed<8>), np, 256, w.plist->as<IngPage>(), w.o_values->as<uint8_t>(),
                           w.o_valid->as<uint32_t>(), page_bufs[c]->as<uint8_t>());
            n_pages[c] = np;
            dev_pages[c] = page_bufs[c]->p;
        } else {
            vpages[c] = ctx->buf((size_t)np * PAGE_BYTES);
            RJ_ILAUNCH(L, "ingest_pages", k_ing_pages_vc, np, 256, w.plist->as<IngPage>(), t, fend->as<uint32_t>(), (uint32_t)n_cols,
                       (uint32_t)c, w.o_row->as<uint32_t>(), w.o_len->as<uint32_t>(), vpages[c]->as<uint8_t>());
            // VARCHAR pages live on the host side of a resident table (the executor resolves strings
            // at the root, rj_varchar*.{cpp,hip})
            TableColumn& tc = tab->cols[c];
            tc.n_pages = np;
            tc.host_pages.resize((size_t)np * PAGE_BYTES);
            RJ_HIP(hipMemcpyAsync(tc.host_pages.data(), vpages[c]->p, (size_t)np * PAGE_BYTES, hipMemcpyDeviceToHost, ctx->stream));
            tc.vc_pages.resize(np);
            for (uint32_t p = 0; p < np; ++p) tc.vc_pages[p] = tc.host_pages.data() + (size_t)p * PAGE_BYTES;
            // ... and stay in HBM too: the device VARCHAR paths (large root results, VARCHAR join keys)
            // then upload nothing for this column
            tc.vc_dev = vpages[c];
        }
    }
    ctx->sync();  // (the columns' work buffers go back to the block cache when this function returns)
    // fixed-width columns: adopt the page images (regularity, row counts: rj_table.hip)
    std::unique_ptr<Table> adopted(table_adopt(ctx, n_out, n_cols, col_type, dev_pages.data(), n_pages.data()));
    for (uint64_t c = 0; c < n_cols; ++c) {
        if (col_type[c] == RJ_VARCHAR) continue;
        tab->cols[c] = std::move(adopted->cols[c]);
        tab->cols[c].owned = page_bufs[c];
    }
    return tab.release();
}

// the device's FP64 field parser, compiled for the host (rj_debug_parse_fp64: CPU tests)
int parse_fp64_host(const char* s, uint64_t n, uint64_t* bits) {
    struct It {
        const char *p, *e;
        bool        next(uint8_t& c) {
            if (p == e) return false;
            c = (uint8_t)*p++;
            return true;
        }
    } it{s, s + n};
    return parse_fp64_field(it, h_pow5, bits);
}

uint64_t table_col_pages(const Table* t, uint64_t col) {
    if (!t || col >= t->cols.size()) return 0;
    const TableColumn& c = t->cols[col];
    return c.type == RJ_VARCHAR ? c.vc_pages.size() : c.n_pages;
}

void table_copy_pages(Context* ctx, const Table* t, uint64_t col, void* const* dst, uint64_t n_dst) {
    if (!t || col >= t->cols.size()) throw_fmt(RJ_ERR_ARG, "column out of range");
    const TableColumn& c = t->cols[col];
    const uint64_t     np = table_col_pages(t, col);
    if (n_dst < np) throw_fmt(RJ_ERR_ARG, "destination has too few pages");
    if (c.type == RJ_VARCHAR) {
        for (uint64_t p = 0; p < np; ++p) memcpy(dst[p], c.vc_pages[p], PAGE_BYTES);
        return;
    }
    if (c.skipped) throw_fmt(RJ_ERR_ARG, "column was not uploaded");
    for (uint64_t p = 0; p < np; ++p)
        RJ_HIP(hipMemcpyAsync(dst[p], c.dev_pages + p * PAGE_BYTES, PAGE_BYTES, hipMemcpyDeviceToHost, ctx->stream));
    ctx->sync();
}

}  // namespace rj
