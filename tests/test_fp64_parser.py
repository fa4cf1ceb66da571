"""The FP64 field parser of the device ingest (csrc/rj_fp64.hpp: Eisel-Lemire), compiled for the
host and called through rj_debug_parse_fp64 — no GPU: whenever it decides a field, the bits are
those of the correctly rounded double (Python's float() is a correctly rounding reader, as
std::from_chars is: reference src/build_table.cpp:57-64); what it leaves undecided is what the
ingest hands to the host's std::from_chars; out-of-range text is reported as the reference's
"parse float error" case (result_out_of_range)."""
import struct

import numpy as np

import _csvgen as g
from pyrj import capi


def bits_of(x: float) -> int:
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def test_decided_fields_are_correctly_rounded():
    rng = np.random.default_rng(2024)
    undecided = 0
    n = 60000
    for _ in range(n):
        t = g.random_fp64_text(rng)
        st, b = capi.parse_fp64(t.encode())
        if st == 2:
            undecided += 1
            continue
        assert st == 0, (t, st)
        assert b == bits_of(float(t)), (t, hex(b), hex(bits_of(float(t))))
    # only texts cut off next to a rounding boundary, or longer than 19 digits and near one, may be left over
    assert undecided < n // 8, undecided


def test_known_boundaries():
    cases = {
        "9007199254740993": 9007199254740992.0,            # 2^53 + 1: a tie, to even
        "9007199254740995": 9007199254740996.0,
        "1.7976931348623157e308": 1.7976931348623157e308,  # the largest double
        "4.9e-324": 5e-324, "5e-324": 5e-324, "3e-324": 5e-324,
        "2.2250738585072014e-308": 2.2250738585072014e-308,  # the smallest normal
        "2.2250738585072011e-308": 2.225073858507201e-308,   # the largest subnormal
        "0.1": 0.1, "-0": -0.0, "0e999": 0.0, "-0.0e-999": -0.0, "1e23": 1e23, "8.41e21": 8.41e21,
        "123456789012345678901234567890": 1.2345678901234568e29,
        "0.000000000000000000000000000000000000000000000000000000000000001": 1e-63,
    }
    for t, want in cases.items():
        st, b = capi.parse_fp64(t.encode())
        assert st == 0 and b == bits_of(want), (t, st, hex(b), hex(bits_of(want)))


def test_out_of_range_and_left_to_the_host():
    for t in ("1e309", "1.7976931348623159e308", "-1e400", "1e-400", "2.4703282292062327e-324", "0." + "0" * 400 + "1", "1" + "0" * 400):
        assert capi.parse_fp64(t.encode())[0] == 1, t
    # not the plain grammar: std::from_chars takes the longest numeric prefix, "inf", "nan" — on the host
    for t in ("inf", "-Infinity", "nan", "nan(1)", "12abc", "1e", "1e+", "0x10", "1.5.2", "+5", ".", "-", "e5", " 1", "1 "):
        assert capi.parse_fp64(t.encode())[0] == 2, t
