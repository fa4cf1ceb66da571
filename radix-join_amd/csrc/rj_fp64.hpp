// rj_fp64.hpp — decimal text -> IEEE double, correctly rounded, for host and device code.
//
// The reference parses FP64 fields with std::from_chars(const char*, const char*, double&)
// (src/build_table.cpp:57-64): the result is the double NEAREST to the decimal value (ties to
// even).  This is the Eisel-Lemire algorithm (D. Lemire, "Number Parsing at a Gigabyte per
// Second", 2021): the decimal significand w (up to 19 digits in 64 bits) times a 128-bit
// approximation of 5^q gives the binary significand with enough spare bits to round in all but a
// vanishing share of the inputs — those, and everything outside the plain number grammar, are
// reported as "undecided" and go to the host's std::from_chars (rj_ingest.hip).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RJ_HD __host__ __device__ __forceinline__
#else
#define RJ_HD inline
#endif

namespace rj {

constexpr int FP64_Q_MIN = -342, FP64_Q_MAX = 308;  // the table's range of decimal exponents

struct U128 {
    uint64_t lo, hi;
};
RJ_HD U128 mul_64x64(uint64_t a, uint64_t b) {
    const unsigned __int128 p = (unsigned __int128)a * b;
    return U128{(uint64_t)p, (uint64_t)(p >> 64)};
}

// w * 10^q -> bits of the nearest double (sign not included).  Returns false when the spare bits
// cannot tell which way to round (the caller falls back); `pow5` is the table of
// rj_pow5_table.inc ({high, low} per exponent).
RJ_HD bool decimal_to_double_bits(uint64_t w, int32_t q, const uint64_t* pow5, uint64_t* bits) {
    if (w == 0 || q < FP64_Q_MIN) {  // zero, or below the smallest subnormal's half
        *bits = 0;
        return true;
    }
    if (q > FP64_Q_MAX) {
        *bits = 0x7ff0000000000000ull;
        return true;
    }
    const int lz = __builtin_clzll(w);
    w <<= lz;
    const uint64_t* t = pow5 + 2 * (q - FP64_Q_MIN);
    U128 prod = mul_64x64(w, t[0]);
    if ((prod.hi & 0x1ffu) == 0x1ffu) {  // the 9 bits below the 55 we keep are all ones: refine
        const U128 second = mul_64x64(w, t[1]);
        prod.lo += second.hi;
        if (second.hi > prod.lo) ++prod.hi;
    }
    if (prod.lo == ~0ull && (q < -27 || q > 55)) return false;
    const int upper = (int)(prod.hi >> 63);
    const int shift = upper + 64 - 52 - 3;
    uint64_t  m = prod.hi >> shift;
    // floor(q * log2(10)) + 63, as (q * 217706) >> 16
    int32_t e2 = (int32_t)((((int64_t)q * 217706) >> 16) + 63) + upper - lz + 1023;
    if (e2 <= 0) {  // subnormal
        if (-e2 + 1 >= 64) {
            *bits = 0;
            return true;
        }
        m >>= -e2 + 1;
        m += m & 1u;
        m >>= 1;
        // rounding may have produced the smallest normal number
        *bits = m;  // (bit 52 set then reads as exponent 1)
        return true;
    }
    // exactly half-way between two doubles: to even
    if (prod.lo <= 1 && q >= -4 && q <= 23 && (m & 3u) == 1u && (m << shift) == prod.hi) m &= ~1ull;
    m += m & 1u;
    m >>= 1;
    if (m >= (2ull << 52)) {
        m = 1ull << 52;
        ++e2;
    }
    m &= ~(1ull << 52);
    if (e2 >= 0x7ff) {
        *bits = 0x7ff0000000000000ull;
        return true;
    }
    *bits = m | ((uint64_t)e2 << 52);
    return true;
}

// What came out of a field: PARSED (value holds the bits), RANGE (the reference's
// result_out_of_range: a finite text whose value is not representable — overflow, or a non-zero
// value that rounds to zero), UNDECIDED (not the plain grammar, or the rounding could not be
// settled: the caller asks the host's std::from_chars).
enum { FP64_PARSED = 0, FP64_RANGE = 1, FP64_UNDECIDED = 2 };

// The plain grammar: ['-'] digits ['.' digits] [('e'|'E') ['+'|'-'] digits], at least one digit in
// the significand, the WHOLE field consumed.  `It` yields the field's characters:
// bool next(uint8_t&).
template <class It>
RJ_HD int parse_fp64_field(It& it, const uint64_t* pow5, uint64_t* bits) {
    uint8_t c;
    if (!it.next(c)) return FP64_UNDECIDED;
    bool neg = false;
    if (c == '-') {
        neg = true;
        if (!it.next(c)) return FP64_UNDECIDED;
    }
    uint64_t w = 0;
    int32_t  nd = 0, q = 0;
    bool     any = false, dropped = false, more = true, frac = false;
    for (;;) {
        if (c >= '0' && c <= '9') {
            any = true;
            const uint32_t d = c - '0';
            if (nd == 0 && d == 0) {
                if (frac) --q;  // a zero right behind the point only moves the scale
            } else if (nd < 19) {
                w = w * 10 + d;
                ++nd;
                if (frac) --q;
            } else {
                dropped |= d != 0;
                if (!frac) ++q;
            }
        } else if (c == '.' && !frac) {
            frac = true;
        } else {
            break;
        }
        if (!it.next(c)) {
            more = false;
            break;
        }
    }
    if (!any) return FP64_UNDECIDED;
    if (more) {  // an exponent, or something the plain grammar does not know
        if (c != 'e' && c != 'E') return FP64_UNDECIDED;
        if (!it.next(c)) return FP64_UNDECIDED;
        bool eneg = false;
        if (c == '+' || c == '-') {
            eneg = c == '-';
            if (!it.next(c)) return FP64_UNDECIDED;
        }
        int32_t e = 0;
        bool    edig = false;
        for (;;) {
            if (c < '0' || c > '9') return FP64_UNDECIDED;
            edig = true;
            if (e < 100000) e = e * 10 + (int32_t)(c - '0');
            if (!it.next(c)) break;
        }
        if (!edig) return FP64_UNDECIDED;
        q += eneg ? -e : e;
    }
    uint64_t b = 0;
    if (!decimal_to_double_bits(w, q, pow5, &b)) return FP64_UNDECIDED;
    if (dropped) {  // more than 19 digits: the true value lies between w and w + 1
        uint64_t b1 = 0;
        if (!decimal_to_double_bits(w + 1, q, pow5, &b1) || b1 != b) return FP64_UNDECIDED;
    }
    if (w != 0 && (b == 0 || b == 0x7ff0000000000000ull)) return FP64_RANGE;
    *bits = b | ((uint64_t)neg << 63);
    return FP64_PARSED;
}

}  // namespace rj
