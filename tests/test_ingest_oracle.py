"""The oracle's restatement of the ingest path (oracle/rjo_ingest.c: CSVParser, TableParser::on_field,
filter bitmaps, ColumnInserter) against independent expectations — no GPU:
  * hand-written CSV with every feature of the dialect (quotes, backslash escapes, CRLF / CR / LF,
    empty = NULL, quoted numbers, no final newline) decodes to the rows a reader of
    src/csv_parser.cpp:3-175 expects;
  * random tables written with random quoting come back exactly, filtered by random postfix
    programs (NULL semantics of the reference's bitmap arithmetic included);
  * the page-fill rule in closed form: 1984 / 1007 values per NULL-free page, and every page the
    inserter closes is FULL (the row behind it would not have fitted, plan.h:204-222,302-331);
  * the reference's errors.
The reference holds no CSV fixture (IMDB is downloaded): beyond these rules the restatement is
"parity unpinned"."""
import numpy as np
import pytest

import _csvgen as g
import _oracle
from pyrj import pages as pg
from pyrj import plan as pl

I32, I64, F64, VC = g.INT32, g.INT64, g.FP64, g.VARCHAR


def clobbered_cells(t: pl.ColumnarTable):
    """A quirk of the reference the restatement keeps: ColumnInserter<T>::insert tests
    `data_end + 4 + num_rows / 8 + 1 > PAGE_SIZE` with a literal 4 (include/plan.h:205), so for an
    8-byte type the last value of a page may reach up to 4 bytes into the validity bitmap, which
    save_page then copies over it (:186).  -> {(row, col)} of the values that read back altered
    (NULL-bearing INT64 / FP64 columns only; IMDB has none)."""
    out = set()
    for ci, c in enumerate(t.columns):
        if c.type not in (I64, F64):
            continue
        r = 0
        for p in c.pages:
            nr, nv = int(p[0]) | int(p[1]) << 8, int(p[2]) | int(p[3]) << 8
            if 8 + 8 * nv > 8192 - (nr + 7) // 8:
                assert 8 + 8 * nv - (8192 - (nr + 7) // 8) <= 4
                bits = np.unpackbits(p[8192 - (nr + 7) // 8:], bitorder="little")[:nr]
                out.add((r + int(np.nonzero(bits)[0][-1]), ci))
            r += nr
    return out


def bitwise(rows):
    """expected rows with every double replaced by ("f64", its bits), as rows_of() reports them"""
    import struct

    return [tuple(("f64", struct.unpack("<Q", struct.pack("<d", float(v)))[0]) if isinstance(v, float) else v for v in row) for row in rows]


def rows_of(t: pl.ColumnarTable, expect=None):
    """decoded rows; with `expect`, the cells clobbered_cells() names take the expected value"""
    cols = []
    for c in t.columns:
        if c.type == VC:
            cols.append(pg.unpack_varchar(c.pages, t.num_rows))
        else:
            v, m = pg.unpack_fixed(c.pages, t.num_rows, c.type)
            if c.type == F64:  # doubles compare by their bits (-0.0 is not 0.0 here)
                b = np.asarray(v).view(np.uint64)
                cols.append([("f64", int(b[i])) if m[i] else None for i in range(t.num_rows)])
            else:
                cols.append([int(v[i]) if m[i] else None for i in range(t.num_rows)])
    if expect is not None:
        for r, ci in clobbered_cells(t):
            cols[ci][r] = expect[r][ci]
    return [tuple(c[i] for c in cols) for i in range(t.num_rows)]


def test_dialect_by_hand():
    text = (
        b'1,plain,10\n'
        b'2,"quoted, with comma",\r\n'                       # empty last field = NULL, CRLF
        b'3,"esc \\" quote and \\\\ backslash and \\n stays",30\r'  # lone CR ends a record; \n after a backslash is literal text
        b'"4",back\\slash outside quotes,-40\n'                # quoted number; backslash outside quotes is literal
        b',"",0\n'                                             # NULL key, NULL string ("" is empty), 0
        b'6,"line\nbreak inside",60\n'
        b'7,"a""b",70'                                         # adjacent quotes just toggle: a + b; no final newline
    )
    t = _oracle.from_csv(text, [I32, VC, I32])
    assert rows_of(t) == [
        (1, b"plain", 10),
        (2, b"quoted, with comma", None),
        (3, b'esc " quote and \\ backslash and \\n stays', 30),
        (4, b"back\\slash outside quotes", -40),
        (None, None, 0),
        (6, b"line\nbreak inside", 60),
        (7, b"ab", 70),
    ]


@pytest.mark.parametrize("seed", range(12))
def test_random_tables_roundtrip_and_filters(seed):
    rng = np.random.default_rng(seed)
    types = [[I32, VC, I64], [VC, I32], [I32, I32, VC, VC], [I64], [F64, I32], [VC, F64, F64]][seed % 6]
    n = int(rng.integers(1, 4000))
    rows = g.random_rows(rng, n, types, null_p=0.15, long_p=0.002 if seed % 3 == 0 else 0.0)
    if len(types) == 1:  # a one-column table cannot tell a NULL row from an empty line: keep both legal
        rows = [r for r in rows if r[0] is not None] or [(1,)]
    text = g.to_csv(rng, rows, final_newline=bool(seed & 1))
    assert rows_of(_oracle.from_csv(text, types), bitwise(rows)) == bitwise(rows)
    for _ in range(4):
        prog = g.random_filter(rng, rows, types)
        want = [row for r, row in enumerate(rows) if g.eval_filter(prog, row, r)]
        got = _oracle.from_csv(text, types, prog)
        assert got.num_rows == len(want)
        assert rows_of(got, bitwise(want)) == bitwise(want)


def test_fp64_fields_by_hand():
    """std::from_chars(double): the longest numeric prefix, no '+', inf / nan in any case, the nearest
    double; an empty field is NULL"""
    text = b'1.5\n-0\n"2.5e3"\n\n.5\n5.\n12abc\n1e\n0x10\ninf\n-Infinity\n9007199254740993\n4.9e-324\n1e-320\n1.5.2\n'
    t = _oracle.from_csv(text + b"7\n", [F64])  # (a last row, so that the empty line before is a row of its own)
    got = rows_of(t)
    want = [1.5, -0.0, 2500.0, None, 0.5, 5.0, 12.0, 1.0, 0.0, float("inf"), float("-inf"), 9007199254740992.0, 5e-324, 1e-320, 1.5, 7.0]
    assert got == bitwise([(w,) for w in want])
    v, m = pg.unpack_fixed(_oracle.from_csv(b"nan\n-nan\nNaN(7)x\n", [F64]).columns[0].pages, 3, F64)
    assert m.all() and np.isnan(v).all()


def _page_shapes(col: pl.Column):
    """(rows, values) of every page; long-string pages as ('L', n_chars)"""
    out = []
    for p in col.pages:
        nr, nv = int(p[0]) | int(p[1]) << 8, int(p[2]) | int(p[3]) << 8
        out.append(("L" if nr == 0xFFFF else "Lc", nv) if nr >= 0xFFFE else (nr, nv))  # first / further page of a long string
    return out


def test_fill_rule_closed_forms():
    n = 5000
    text = b"".join(b"%d,%d\n" % (i, i * 3) for i in range(n))
    t = _oracle.from_csv(text, [I32, I64])
    assert _page_shapes(t.columns[0]) == [(1984, 1984), (1984, 1984), (1032, 1032)]
    assert _page_shapes(t.columns[1]) == [(1007, 1007)] * 4 + [(972, 972)]
    # all NULL: a row costs one bitmap bit — ColumnInserter<T>::insert_null closes the page when
    # data_end + num_rows / 8 + 1 > 8192 (plan.h:217)
    t = _oracle.from_csv(b"\n" * 70000, [I32])  # (one column: an empty line is a NULL row)
    shapes = _page_shapes(t.columns[0])
    assert sum(s[0] for s in shapes) == 70000 and all(s[1] == 0 for s in shapes)
    assert shapes[0][0] == (8192 - 4) * 8  # 4 + nr / 8 + 1 <= 8192  <=>  nr <= 65503: 65504 rows fit


@pytest.mark.parametrize("seed", range(6))
def test_every_closed_page_is_full(seed):
    rng = np.random.default_rng(100 + seed)
    types = [I32, I64, VC]
    n = 30000
    rows = g.random_rows(rng, n, types, null_p=[0.0, 0.3, 0.9][seed % 3], long_p=0.0005)
    t = _oracle.from_csv(g.to_csv(rng, rows), types)
    assert rows_of(t, rows) == rows
    for c, ty in enumerate(types):
        r = 0
        shapes = _page_shapes(t.columns[c])
        for k, (nr, nv) in enumerate(shapes):
            if nr in ("L", "Lc"):
                r += nr == "L"  # a long string is ONE row, whatever pages it takes
                continue
            last = k + 1 == len(shapes)
            nxt = rows[r + nr][c] if r + nr < n else None
            if ty == VC:
                chars = sum(len(rows[i][c]) for i in range(r, r + nr) if rows[i][c] is not None)
                used = 4 + 2 * nv + chars
                assert used + (nr - 1) // 8 + 1 <= 8192
                if not last and not (nxt is not None and len(nxt) > 8185):
                    extra = 0 if nxt is None else 2 + len(nxt)
                    assert used + extra + nr // 8 + 1 > 8192  # the next row did not fit (plan.h:307,326)
            else:
                w = 4 if ty == I32 else 8
                used = w + nv * w
                assert used + (nr + 7) // 8 <= 8192 + (4 if ty == I64 else 0)  # (see clobbered_cells)
                if not last:
                    assert used + (4 if nxt is not None else 0) + nr // 8 + 1 > 8192  # plan.h:205,217
            r += nr
        # long strings take pages of their own: one row per chain
        n_long = sum(1 for row in rows if row[c] is not None and ty == VC and len(row[c]) > 8185)
        assert sum(s[0] for s in shapes if s[0] not in ("L", "Lc")) + n_long == n
        assert sum(1 for s in shapes if s[0] == "L") == n_long


@pytest.mark.parametrize("text,types,msg", [
    (b"1,2\n3\n", [I32, I32], "CSV parse error"),              # a record with fewer fields
    (b"1,2,3\n", [I32, I32], "CSV parse error"),               # ... with more
    (b'1,"open\n', [I32, VC], "CSV parse error"),              # QuoteNotClosed
    (b"1,x\n", [I32, I32], "parse integer error"),
    (b"1,-\n", [I32, I32], "parse integer error"),
    (b"1,2147483648\n", [I32, I32], "parse integer error"),    # out of range for INT32
    (b"1,9223372036854775808\n", [I32, I64], "parse integer error"),
    (b"1,x\n", [I32, F64], "parse float error"),
    (b"1,+5\n", [I32, F64], "parse float error"),               # std::from_chars takes no '+'
    (b"1, 5\n", [I32, F64], "parse float error"),               # ... and skips no white space
    (b"1,.\n", [I32, F64], "parse float error"),
    (b"1,1e999\n", [I32, F64], "parse float error"),            # result_out_of_range: overflow
    (b"1,1e-999\n", [I32, F64], "parse float error"),           # ... and non-zero text that rounds to zero
])
def test_errors_as_the_reference_raises_them(text, types, msg):
    with pytest.raises(RuntimeError) as e:
        _oracle.from_csv(text, types)
    assert msg in str(e.value)


def test_string_comparisons_follow_std_string():
    text = b'abc\n"ab"\n\nabd\n"a,b"\n\xc3\xa9\n'
    t = _oracle.from_csv(text, [VC], [("GEQ", 0, b"ab"), ("LT", 0, b"abd"), ("AND",)])
    assert rows_of(t) == [(b"abc",), (b"ab",)]
    t = _oracle.from_csv(text, [VC], [("GT", 0, b"abd")])  # bytes compare unsigned: 0xc3 > 'a'
    assert rows_of(t) == [(b"\xc3\xa9",)]
    t = _oracle.from_csv(text, [VC], [("NEQ", 0, b"abc")])  # false on NULL (bitmap & cmp)
    assert len(rows_of(t)) == 4


def test_like_is_what_re2_is_asked_for():
    """'%' any run, '_' any ONE character — a UTF-8 character, never a newline; the whole string must
    match; NULL is false for LIKE and for NOT LIKE; ill-formed UTF-8 matches nothing (so NOT LIKE
    holds for it)"""
    rows = [b"abc", b"abcd", b"xbc", None, b"a\nc", "\u00e9t\u00e9".encode(), b"a\xffc", b"%", b"a.c", b"ac"]
    text = b"".join((b'"' + r.replace(b'"', b'\\"') + b'"' if r is not None else b"") + b"\n" for r in rows)
    def run(prog):
        return [r[0] for r in rows_of(_oracle.from_csv(text, [VC], prog))]
    assert run([("LIKE", 0, b"a%")]) == [b"abc", b"abcd", b"a.c", b"ac"]  # 'a\nc': '.' stops at the newline; 'a\xffc': ill-formed
    assert run([("LIKE", 0, b"a_c")]) == [b"abc", b"a.c"]
    assert run([("LIKE", 0, b"_bc%")]) == [b"abc", b"abcd", b"xbc"]
    assert run([("LIKE", 0, "_t_".encode())]) == ["\u00e9t\u00e9".encode()]  # '_' is one CHARACTER, two bytes here
    assert run([("LIKE", 0, "\u00e9%".encode())]) == ["\u00e9t\u00e9".encode()]
    assert run([("LIKE", 0, b"a.c")]) == [b"a.c"]  # regex metacharacters are literals
    assert run([("NOT_LIKE", 0, b"a%")]) == [b"xbc", b"a\nc", "\u00e9t\u00e9".encode(), b"a\xffc", b"%"]  # not the NULL row
    assert run([("LIKE", 0, b"\xff%")]) == []  # a pattern that is not UTF-8 does not compile: false ...
    assert len(run([("NOT_LIKE", 0, b"\xff%")])) == 9  # ... so NOT LIKE holds for every non-NULL row
    assert run([("LIKE", 0, b"%")]) == [r for r in rows if r is not None and b"\n" not in r and b"\xff" not in r]
    for r in rows:
        for pat in (b"a%", b"%c", b"a_c", b"%b%", b"_%_", b"abc", b"a%c%", b"%%a", b"_"):
            if r is not None:
                got = run([("LIKE", 0, pat)])
                assert (r in got) == g.like_model(r, pat), (r, pat)


def test_from_chars_takes_a_prefix_and_the_int32_literal_is_truncated():
    t = _oracle.from_csv(b"12abc,-2147483648\n7,2147483647\n", [I32, I32])  # trailing garbage is ignored (std::from_chars)
    assert rows_of(t) == [(12, -(2**31)), (7, 2**31 - 1)]
    # Comparison::eval casts the literal to int32_t for an INT32 column (statement.cpp:55): 2^32 + 7 == 7
    t = _oracle.from_csv(b"12,1\n7,2\n", [I32, I32], [("EQ", 0, 2**32 + 7)])
    assert rows_of(t) == [(7, 2)]
