"""Wide carries (CARRY_WIDE): when a join side must deliver several non-key columns, or a column
with NULLs, their VALUES travel with the key through the radix passes — up to MAX_WORDS - KW carry
words: two or three 32-bit words, or a 64-bit column plus one 32-bit word; validity bits ride in a
word of their own — instead of a row index that every column is gathered through afterwards
(reference semantics: src/execute.cpp:236-242 copies any column list per output row).  Every
record layout, on either side, behind the broadcast join, one pass and two passes, as root and as
the input of a parent join, against the oracle."""
import numpy as np
import pytest

import _oracle
from pyrj import capi
from pyrj import plan as pl

pytestmark = pytest.mark.gpu

I32, I64, F64, VC = pl.INT32, pl.INT64, pl.FP64, pl.VARCHAR
NP_OF = {I32: np.int32, I64: np.int64, F64: np.float64}

# (key type, payload columns of the wide side as (type, nullable))
LAYOUTS = {
    "32_32": (I32, [(I32, False), (I32, False)]),
    "32_32_32": (I32, [(I32, False), (I32, False), (I32, False)]),
    "64_32": (I32, [(I32, False), (F64, False)]),            # 64-bit column first in the record, whatever the plan's order
    "32_V": (I32, [(I32, True)]),
    "32_32_V": (I32, [(I32, True), (I32, False)]),
    "64_V": (I32, [(I64, True)]),
    "k64_32_32": (I64, [(I32, False), (I32, False)]),
    "k64_32_V": (F64, [(I32, True)]),
    "vc_32": (I32, [(VC, True), (I32, False)]),              # a VARCHAR column travels as its 32-bit row id
}


def column(rng, n, dt, nullable):
    if dt == VC:
        v = [f"s{int(x)}".encode() * (1 + int(x) % 3) for x in rng.integers(0, 5000, n)]
        return (dt, v, rng.random(n) >= 0.2) if nullable else (dt, v)
    if dt == F64:
        v = rng.standard_normal(n)
    elif dt == I64:
        v = rng.integers(-(2**62), 2**62, n)
    else:
        v = rng.integers(-(2**31), 2**31 - 1, n)
    v = v.astype(NP_OF[dt])
    return (dt, v, rng.random(n) >= 0.25) if nullable else (dt, v)


def keys(rng, n, domain, kt):
    k = rng.integers(0, domain, n)
    if kt == F64:
        return k.astype(np.float64) * 0.25 - 3.0
    if kt == I64:
        return k.astype(np.int64) * 3_000_000_019 - 5
    return k.astype(np.int32) - 11


@pytest.fixture(scope="module", params=[0, 9, 16], ids=["auto", "one_pass", "two_passes"])
def ctx(request):
    c = capi.Context(radix_bits=request.param)
    yield c
    capi.destroy_context(c)


@pytest.mark.parametrize("wide_is_build", [True, False])
@pytest.mark.parametrize("layout", list(LAYOUTS))
def test_wide_layouts(ctx, layout, wide_is_build):
    kt, pay = LAYOUTS[layout]
    rng = np.random.default_rng(abs(hash((layout, wide_is_build))) % 2**31)
    n_wide, n_other, dom = (120_000, 260_000, 90_000) if layout != "vc_32" else (30_000, 60_000, 22_000)
    wide_cols = [(kt, keys(rng, n_wide, dom, kt), rng.random(n_wide) >= 0.03)] + [column(rng, n_wide, dt, nl) for dt, nl in pay]
    other_cols = [(kt, keys(rng, n_other, dom + 7000, kt)), column(rng, n_other, I32, False)]
    wt, ot = pl.make_table(wide_cols), pl.make_table(other_cols)
    p = pl.Plan()
    w = p.new_scan_node(0, [(i, c[0]) for i, c in enumerate(wide_cols)])
    o = p.new_scan_node(1, [(i, c[0]) for i, c in enumerate(other_cols)])
    nw = len(wide_cols)
    # left = the wide side; every payload column out (in reverse order), the key, the other side's payload
    outs = [(i, wide_cols[i][0]) for i in range(nw - 1, 0, -1)] + [(0, kt), (nw + 1, I32)]
    j = p.new_join_node(wide_is_build, w, o, 0, 0, outs)
    p.new_input(wt)
    p.new_input(ot)
    p.root = j
    got, want = capi.execute(p, ctx), _oracle.execute(p)
    assert got.num_rows == want.num_rows
    assert [c.type for c in got.columns] == [c.type for c in want.columns]
    if layout == "vc_32":  # (the digest covers fixed-width columns only)
        assert pl.canonical_rows(got) == pl.canonical_rows(want)
    else:
        assert pl.table_digest(got) == pl.table_digest(want)


def test_wide_result_feeds_a_parent_join(ctx):
    """(A ⋈ B) ⋈ C: the first join's output columns (split out of its records, validity included)
    are the second join's key and payload columns; wide carries on both levels."""
    rng = np.random.default_rng(77)
    na, nb, nc = 150_000, 300_000, 40_000
    a = pl.make_table([(I32, rng.integers(0, 100_000, na).astype(np.int32)), column(rng, na, I32, True), column(rng, na, I64, False)])
    b = pl.make_table([(I32, rng.integers(0, 110_000, nb).astype(np.int32)), (I32, rng.integers(0, 30_000, nb).astype(np.int32), rng.random(nb) > 0.1), column(rng, nb, I32, False)])
    c = pl.make_table([(I32, rng.integers(0, 30_000, nc).astype(np.int32)), column(rng, nc, F64, True)])
    p = pl.Plan()
    sa = p.new_scan_node(0, [(0, I32), (1, I32), (2, I64)])
    sb = p.new_scan_node(1, [(0, I32), (1, I32), (2, I32)])
    j1 = p.new_join_node(True, sa, sb, 0, 0, [(1, I32), (2, I64), (4, I32), (5, I32)])  # a.x(nullable), a.y, b.fk(nullable), b.z
    sc = p.new_scan_node(2, [(0, I32), (1, F64)])
    j2 = p.new_join_node(False, j1, sc, 2, 0, [(0, I32), (5, F64), (1, I64), (3, I32), (2, I32)])
    for t in (a, b, c):
        p.new_input(t)
    p.root = j2
    got, want = capi.execute(p, ctx), _oracle.execute(p)
    assert got.num_rows == want.num_rows
    assert pl.table_digest(got) == pl.table_digest(want)


def test_row_index_carry_still_serves_what_does_not_fit():
    """Four payload words on one side do not fit behind the key: that side falls back to a row
    index + gathers, the other side's two columns still travel with the key."""
    rng = np.random.default_rng(78)
    nb, npr = 80_000, 200_000
    bt = pl.make_table([(I32, rng.permutation(nb).astype(np.int32)), column(rng, nb, I64, False), column(rng, nb, I64, True)])
    pt = pl.make_table([(I32, rng.integers(0, nb + 500, npr).astype(np.int32)), column(rng, npr, I32, False), column(rng, npr, I32, False)])
    p = pl.Plan()
    p.new_scan_node(0, [(0, I32), (1, I64), (2, I64)])
    p.new_scan_node(1, [(0, I32), (1, I32), (2, I32)])
    p.new_join_node(True, 0, 1, 0, 0, [(1, I64), (2, I64), (4, I32), (5, I32), (3, I32)])
    p.new_input(bt)
    p.new_input(pt)
    p.root = 2
    for bits in (0, 12):
        c = capi.Context(radix_bits=bits)
        try:
            got = capi.execute(p, c)
        finally:
            c.destroy()
        want = _oracle.execute(p)
        assert got.num_rows == want.num_rows and pl.table_digest(got) == pl.table_digest(want)
