"""Table::from_csv on the device (rj_table_from_csv, csrc/rj_ingest.hip) against
the oracle's restatement of the reference path (oracle/rjo_ingest.c): the resident table's pages
must equal the oracle's BYTE FOR BYTE — same rows, same page boundaries (ColumnInserter's fill
rule), same headers, offsets, characters and bitmaps — for random tables written with random
quoting, escapes, CRLF / CR / LF, NULLs and long strings, under random filter programs (numeric
comparisons and IS [NOT] NULL on the device, string predicates as host bitmaps); the reference's
errors; and a join over two tables ingested on the device."""
import numpy as np
import pytest

import _csvgen as g
import _oracle
from pyrj import capi
from pyrj import plan as pl

pytestmark = pytest.mark.gpu
I32, I64, F64, VC = g.INT32, g.INT64, g.FP64, g.VARCHAR


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context()
    yield c
    capi.destroy_context(c)


def same_pages(ctx, text, types, prog=None):
    want = _oracle.from_csv(text, types, prog)
    t = ctx.from_csv(text, types, prog)
    try:
        got = ctx.table_to_host(t)
    finally:
        t.release()
    assert got.num_rows == want.num_rows
    for c, (a, b) in enumerate(zip(got.columns, want.columns)):
        assert a.type == b.type and a.pages.shape == b.pages.shape, (c, a.pages.shape, b.pages.shape)
        if not np.array_equal(a.pages, b.pages):
            pgi = int(np.nonzero((a.pages != b.pages).any(axis=1))[0][0])
            at = int(np.nonzero(a.pages[pgi] != b.pages[pgi])[0][0])
            raise AssertionError(f"column {c}: page {pgi} differs from byte {at}: {a.pages[pgi][at:at+16]} vs {b.pages[pgi][at:at+16]}")
    return want


def test_dialect_by_hand(ctx):
    text = (b'1,plain,10\n2,"quoted, with comma",\r\n3,"esc \\" quote and \\\\ backslash and \\n stays",30\r'
            b'"4",back\\slash outside quotes,-40\n,"",0\n6,"line\nbreak inside",60\n7,"a""b",70')
    want = same_pages(ctx, text, [I32, VC, I32])
    assert want.num_rows == 7


@pytest.mark.parametrize("seed", range(21))
def test_random_tables_and_filters(ctx, seed):
    rng = np.random.default_rng(7000 + seed)
    types = [[I32, VC, I64], [VC, I32], [I32, I32, VC, VC], [I64, I32], [I32], [F64, I32], [VC, F64, F64, I64]][seed % 7]
    n = int(rng.integers(1, 60000)) if seed % 4 else int(rng.integers(1, 700))
    rows = g.random_rows(rng, n, types, null_p=[0.0, 0.1, 0.5, 0.95][seed % 4], long_p=0.001 if seed % 3 == 0 else 0.0)
    if len(types) == 1:
        rows = [r for r in rows if r[0] is not None] or [(1,)]
    text = g.to_csv(rng, rows, final_newline=bool(seed & 1))
    same_pages(ctx, text, types)
    for _ in range(3):
        same_pages(ctx, text, types, g.random_filter(rng, rows, types))


def test_structure_across_segment_boundaries(ctx):
    """quotes, escapes and CRLF pairs that straddle the 256-byte segments and 64 KiB super-segments
    the parallel parser cuts the text into: fields sized so that every phase occurs"""
    rng = np.random.default_rng(5)
    rows = []
    for k in range(6000):
        body = bytes(rng.integers(97, 123, int(rng.integers(0, 90))).astype(np.uint8))
        tricky = [b'\\"', b"\\\\", b",", b"\r\n", b'"', b"\\", b"\n"][k % 7]
        rows.append((k, body[: len(body) // 2] + tricky + body[len(body) // 2:] or None))
    text = g.to_csv(rng, rows, crlf_p=0.5)
    same_pages(ctx, text, [I32, VC])
    for shift in (1, 2, 3, 129, 255):  # move every boundary through the segment grid
        same_pages(ctx, (b"0," + b"x" * shift + b"\n") + text, [I32, VC])


def test_long_strings_and_all_null_pages(ctx):
    rng = np.random.default_rng(6)
    rows = [(i, None if i % 5 else bytes(rng.integers(32, 127, int(rng.integers(8186, 40000))).astype(np.uint8))) for i in range(60)]
    same_pages(ctx, g.to_csv(rng, rows), [I32, VC])
    same_pages(ctx, b"\n" * 140000, [I32])      # all NULL: 65504 rows per page
    same_pages(ctx, b"\n" * 140000, [VC])
    same_pages(ctx, b",\n" * 70001, [I64, VC])


@pytest.mark.parametrize("text,types,msg", [
    (b"1,2\n3\n", [I32, I32], "CSV parse error"),
    (b"1,2,3\n", [I32, I32], "CSV parse error"),
    (b'1,"open\n', [I32, VC], "CSV parse error"),
    (b"1,x\n", [I32, I32], "parse integer error"),
    (b"1,-\n", [I32, I32], "parse integer error"),
    (b"1,2147483648\n", [I32, I32], "parse integer error"),
    (b"1,9223372036854775808\n", [I32, I64], "parse integer error"),
    (b"1,x\n", [I32, F64], "parse float error"),
    (b"1,+5\n", [I32, F64], "parse float error"),
    (b"1,.\n2,3\n", [I32, F64], "parse float error"),
    (b"1,1e999\n", [I32, F64], "parse float error"),
    (b"1,-1e-999\n", [I32, F64], "parse float error"),
    (b"1,2.4703282292062327e-324\n", [I32, F64], "parse float error"),
    (b"1," + b"9" * 400 + b"\n", [I32, F64], "parse float error"),
])
def test_errors_as_the_reference_raises_them(ctx, text, types, msg):
    with pytest.raises(capi.RjError) as e:
        ctx.from_csv(text, types)
    assert e.value.code == 4 and msg in e.value.message  # RJ_ERR_DATA
    with pytest.raises(RuntimeError) as e2:
        _oracle.from_csv(text, types)
    assert msg in str(e2.value)


def test_prefix_parse_literal_truncation_and_unsupported(ctx):
    same_pages(ctx, b"12abc,-2147483648\n7,2147483647\n", [I32, I32])
    same_pages(ctx, b"12,1\n7,2\n", [I32, I32], [("EQ", 0, 2**32 + 7)])
    same_pages(ctx, b"", [I32, VC])
    same_pages(ctx, b'abc\n"ab"\n\nabd\n"a,b"\nab\\\n', [VC], [("GEQ", 0, b"ab"), ("LT", 0, b"abd"), ("AND",)])
    same_pages(ctx, b'abc\nab\n\n', [VC], [("EQ", 0, b"")])  # nothing equals the empty string: an empty field is NULL
    with pytest.raises(capi.RjError) as e:
        ctx.from_csv(b"1\n", [I32], [("AND",)])
    assert e.value.code == 1


def test_fp64_fields(ctx):
    """std::from_chars(double) (reference src/build_table.cpp:57-64): plain numbers are decided on the
    device, everything else it accepts — the longest numeric prefix, inf / nan in any case, quoted
    fields — by the host's std::from_chars; pages byte for byte the oracle's (NaN payloads included)."""
    text = (b'1.5,1\n-0,2\n"2.5e3",3\n,4\n.5,5\n5.,6\n-.5e-3,7\n12abc,8\n1e,9\n1e+,10\n0x10,11\ninf,12\n-INF,13\n'
            b'Infinity,14\nnan,15\n-nan,16\nNaN(12),17\n1.5.2,18\n9007199254740993,19\n4.9e-324,20\n1.7976931348623157e308,21\n'
            b'0.000000000000000000000000000000000001,22\n123456789012345678901234567890,23\n'
            b'1.00000000000000011102230246251565404236316680908203125,24\n"1.0\\"",25\n')
    want = same_pages(ctx, text, [F64, I32])
    assert want.num_rows == 25
    same_pages(ctx, text, [F64, I32], [("GT", 0, 1.0), ("IS_NULL", 0), ("OR",)])
    same_pages(ctx, text, [F64, I32], [("NEQ", 0, float("nan"))])   # a NaN differs from everything, itself included
    same_pages(ctx, text, [F64, I32], [("EQ", 0, 0.0)])            # -0 equals 0
    rng = np.random.default_rng(77)
    rows = g.random_rows(rng, 300_000, [F64, F64], null_p=0.3)     # NULL-bearing 8-byte pages: the +4 quirk of the inserter
    same_pages(ctx, g.to_csv(rng, rows), [F64, F64])


def test_like_on_the_device(ctx):
    rows = [b"abc", b"abcd", b"xbc", None, b"a\nc", "\u00e9t\u00e9".encode(), b"a\xffc", b"%", b"a.c", b"ac", "\U0001f600x".encode()]
    text = b"".join((b'"' + r.replace(b'"', b'\\"') + b'"' if r is not None else b"") + b"\n" for r in rows) * 50
    for pat in (b"a%", b"a_c", b"_bc%", "_t_".encode(), "\u00e9%".encode(), b"a.c", b"\xff%", b"%", b"%c", b"%b%", b"_%_", b"a%c%", b"%%a", b"_x", b""):
        same_pages(ctx, text, [VC], [("LIKE", 0, pat)])
        same_pages(ctx, text, [VC], [("NOT_LIKE", 0, pat), ("IS_NOT_NULL", 0), ("AND",)])
    with pytest.raises(capi.RjError) as e:
        ctx.from_csv(b"1\n", [I32], [("LIKE", 0, b"1%")])
    assert e.value.code == 1
    with pytest.raises(capi.RjError) as e:
        ctx.from_csv(b"a\n", [VC], [("LIKE", 0, b"a" * 64)])
    assert e.value.code == 5


@pytest.fixture(params=["host_varchar", "device_varchar"])
def jctx(request):
    """two contexts: VARCHAR root columns resolved on the host (small results), and with the device
    path forced (RJ_TUNE_VARCHAR_DEV=1) — which reads the VARCHAR pages the ingest left in HBM"""
    import os

    old = os.environ.get("RJ_TUNE_VARCHAR_DEV")
    if request.param == "device_varchar":
        os.environ["RJ_TUNE_VARCHAR_DEV"] = "1"
    try:
        c = capi.Context()
    finally:
        if request.param == "device_varchar":
            if old is None:
                del os.environ["RJ_TUNE_VARCHAR_DEV"]
            else:
                os.environ["RJ_TUNE_VARCHAR_DEV"] = old
    yield c
    capi.destroy_context(c)


def test_join_over_tables_ingested_on_the_device(jctx):
    """title-like and cast-like tables arrive as CSV text, are filtered and packed on the device, and
    the resident tables go straight into rj_execute_resident: nothing is uploaded at execute() time
    (the VARCHAR pages, too, stay where the ingest wrote them).
    Checked against the oracle end to end (its own from_csv, then its execute)."""
    ctx = jctx
    rng = np.random.default_rng(8)
    nt, nc = 60_000, 250_000
    trows = [(int(i), (b"title %d" % i) if i % 11 else None, int(rng.integers(1900, 2025)) if i % 7 else None) for i in rng.permutation(nt)]
    crows = [(int(rng.integers(0, nt + 5000)), int(rng.integers(0, 5000)) if k % 13 else None, (b"note,%d" % k) if k % 3 else None) for k in range(nc)]
    ttext, ctext = g.to_csv(rng, trows), g.to_csv(rng, crows)
    tfilter = [("GT", 2, 1950), ("IS_NULL", 1), ("NOT",), ("AND",)]
    cfilter = [("IS_NOT_NULL", 1)]
    ttypes, ctypes_ = [I32, VC, I32], [I32, I32, VC]
    p = pl.Plan()
    a = p.new_scan_node(0, [(0, I32), (1, VC), (2, I32)])
    b = p.new_scan_node(1, [(0, I32), (1, I32), (2, VC)])
    p.root = p.new_join_node(True, a, b, 0, 0, [(0, I32), (1, VC), (4, I32), (5, VC), (2, I32)])
    p.new_input(_oracle.from_csv(ttext, ttypes, tfilter))
    p.new_input(_oracle.from_csv(ctext, ctypes_, cfilter))
    want = _oracle.execute(p)
    T, Cc = ctx.from_csv(ttext, ttypes, tfilter), ctx.from_csv(ctext, ctypes_, cfilter)
    try:
        res = ctx.execute_resident(p, [T, Cc])
        got = res.to_table()
        res.free()
    finally:
        T.release()
        Cc.release()
    assert got.num_rows == want.num_rows and got.num_rows > 10_000
    assert pl.canonical_rows(got) == pl.canonical_rows(want)


def test_two_million_rows(ctx):
    """cast_info-sized slice: 2 M rows x (INT32, INT32, VARCHAR, INT32), ~60 MB of text"""
    rng = np.random.default_rng(9)
    n = 2_000_000
    a = rng.integers(0, 4_000_000, n)
    b = rng.integers(0, 1000, n)
    d = rng.integers(1, 12, n)
    lines = [b"%d,%s,\"(as %d, uncredited)\",%d\n" % (a[i], b"" if b[i] < 70 else b"%d" % b[i], b[i], d[i]) for i in range(n)]
    text = b"".join(lines)
    import time

    t0 = time.perf_counter()
    t = ctx.from_csv(text, [I32, I32, VC, I32], [("LT", 3, 5), ("IS_NOT_NULL", 1), ("AND",)])
    dt = time.perf_counter() - t0
    got = ctx.table_to_host(t)
    t.release()
    print(f"from_csv on the device: {len(text) / 1e6:.0f} MB, {n} rows -> {got.num_rows} rows in {dt * 1e3:.0f} ms ({len(text) / dt / 1e9:.2f} GB/s incl. the upload)")
    want = _oracle.from_csv(text, [I32, I32, VC, I32], [("LT", 3, 5), ("IS_NOT_NULL", 1), ("AND",)])
    assert got.num_rows == want.num_rows
    for x, y in zip(got.columns, want.columns):
        assert np.array_equal(x.pages, y.pages)
