"""Edge cases of the join path, GPU vs oracle (bit-exact multisets)."""
import numpy as np
import pytest

import _oracle
from pyrj import capi
from pyrj import pages as pg
from pyrj import plan as pl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.build_context()
    yield c
    capi.destroy_context(c)


def check(ctx, plan, small=True):
    got = capi.execute(plan, ctx)
    want = _oracle.execute(plan)
    assert got.num_rows == want.num_rows
    assert [c.type for c in got.columns] == [c.type for c in want.columns]
    if small:
        assert pl.sorted_rows(got) == pl.sorted_rows(want)
    else:
        assert pl.table_digest(got) == pl.table_digest(want)
    return got


def join2(lt, rt, build_left, la, ra, louts, routs, outs):
    p = pl.Plan()
    p.new_scan_node(0, louts)
    p.new_scan_node(1, routs)
    p.new_join_node(build_left, 0, 1, la, ra, outs)
    p.new_input(lt)
    p.new_input(rt)
    p.root = 2
    return p


def test_all_null_keys_give_an_empty_typed_result(ctx):
    n = 5000
    lt = pl.make_table([(pl.INT32, np.arange(n, dtype=np.int32), np.zeros(n, bool)), (pl.INT32, np.arange(n, dtype=np.int32))])
    rt = pl.make_table([(pl.INT32, np.arange(n, dtype=np.int32)), (pl.INT64, np.arange(n, dtype=np.int64))])
    both32 = [(0, pl.INT32), (1, pl.INT32)]
    for bl in (True, False):
        got = check(ctx, join2(lt, rt, bl, 0, 0, both32, [(0, pl.INT32), (1, pl.INT64)], [(1, pl.INT32), (3, pl.INT64), (0, pl.INT32)]))
        assert got.num_rows == 0 and all(c.pages.shape[0] == 0 for c in got.columns)


def test_self_join_same_pages_twice(ctx):
    rng = np.random.default_rng(1)
    k = rng.integers(0, 3000, 20_000).astype(np.int32)
    t = pl.make_table([(pl.INT32, k), (pl.INT32, np.arange(k.size, dtype=np.int32))])
    both = [(0, pl.INT32), (1, pl.INT32)]
    got = check(ctx, join2(t, t, False, 0, 0, both, both, [(1, pl.INT32), (3, pl.INT32)]), small=False)
    assert got.num_rows == int((np.bincount(k) ** 2).sum())


def test_short_irregular_pages_go_through_page_decode(ctx):
    """non-NULL columns whose pages are NOT full (a legal layout other producers may emit):
    not addressable in place, so K1 decodes them"""
    rng = np.random.default_rng(2)
    n = 30_000
    k = rng.integers(0, 9000, n).astype(np.int32)
    v = rng.integers(-(2**62), 2**62, n).astype(np.int64)

    def short_pages(vals, dtype, rows_per_page):
        parts = [pg.pack_fixed(vals[i : i + rows_per_page], None, dtype) for i in range(0, len(vals), rows_per_page)]
        return np.concatenate(parts)

    lt = pl.ColumnarTable(n, [pl.Column(pl.INT32, short_pages(k, pl.INT32, 700)), pl.Column(pl.INT64, short_pages(v, pl.INT64, 333))])
    rk = rng.permutation(9000).astype(np.int32)
    rt = pl.make_table([(pl.INT32, rk), (pl.INT32, np.arange(9000, dtype=np.int32))])
    check(ctx, join2(lt, rt, False, 0, 0, [(0, pl.INT32), (1, pl.INT64)], [(0, pl.INT32), (1, pl.INT32)],
                     [(1, pl.INT64), (3, pl.INT32), (0, pl.INT32), (2, pl.INT32)]), small=False)


def test_int64_key_with_varchar_and_multi_column_sides(ctx):
    """KW = 2 with a row-index carry (several payload columns per side, one of them VARCHAR)"""
    rng = np.random.default_rng(3)
    nb, npr = 4000, 9000
    bk = rng.integers(-(2**60), 2**60, nb).astype(np.int64)
    pk = rng.choice(bk, npr)
    pk[::7] = rng.integers(-(2**60), 2**60, pk[::7].size)
    bt = pl.make_table([(pl.INT64, bk, rng.random(nb) > 0.05), (pl.VARCHAR, [f"b{i}".encode() if i % 9 else None for i in range(nb)]),
                        (pl.FP64, rng.standard_normal(nb), rng.random(nb) > 0.2)])
    pt = pl.make_table([(pl.INT32, rng.integers(0, 9, npr).astype(np.int32)), (pl.INT64, pk), (pl.INT64, rng.integers(0, 2**40, npr).astype(np.int64))])
    check(ctx, join2(bt, pt, True, 0, 1, [(0, pl.INT64), (1, pl.VARCHAR), (2, pl.FP64)], [(0, pl.INT32), (1, pl.INT64), (2, pl.INT64)],
                     [(1, pl.VARCHAR), (5, pl.INT64), (2, pl.FP64), (3, pl.INT32), (4, pl.INT64), (0, pl.INT64)]))


def test_bushy_tree(ctx):
    """(A ⋈ B) ⋈ (C ⋈ D): both children of the root are joins"""
    rng = np.random.default_rng(4)

    def tab(n, dom):
        return pl.make_table([(pl.INT32, rng.integers(0, dom, n).astype(np.int32), rng.random(n) > 0.03), (pl.INT32, rng.integers(0, 1000, n).astype(np.int32))])

    p = pl.Plan()
    both = [(0, pl.INT32), (1, pl.INT32)]
    a, b = p.new_scan_node(0, both), p.new_scan_node(1, both)
    ab = p.new_join_node(True, a, b, 0, 0, [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)])
    c, d = p.new_scan_node(2, both), p.new_scan_node(3, both)
    cd = p.new_join_node(False, c, d, 0, 0, [(1, pl.INT32), (0, pl.INT32), (3, pl.INT32)])
    root = p.new_join_node(False, ab, cd, 0, 1, [(0, pl.INT32), (1, pl.INT32), (2, pl.INT32), (3, pl.INT32), (5, pl.INT32)])
    for n, dom in ((3000, 800), (2500, 800), (2000, 800), (1500, 800)):
        p.new_input(tab(n, dom))
    p.root = root
    check(ctx, p, small=False)


@pytest.mark.parametrize("bits", [1, 3, 12, 20])
def test_forced_radix_plans(bits):
    """rj_config.radix_bits override: 1-3 bits leave build partitions far larger than the LDS
    table (many table-sized chunks per partition), 12 = two passes on a small input, 20 = three
    passes with mostly empty partitions — results must not depend on the bit plan"""
    c = capi.Context(radix_bits=bits)
    try:
        rng = np.random.default_rng(100 + bits)
        nb, npr = 60_000, 150_000
        bk = rng.integers(0, 50_000, nb).astype(np.int32)
        pk = rng.integers(0, 55_000, npr).astype(np.int32)
        bt = pl.make_table([(pl.INT32, bk, rng.random(nb) > 0.02), (pl.INT32, np.arange(nb, dtype=np.int32))])
        pt = pl.make_table([(pl.INT32, pk), (pl.INT64, np.arange(npr, dtype=np.int64))])
        check(c, join2(bt, pt, True, 0, 0, [(0, pl.INT32), (1, pl.INT32)], [(0, pl.INT32), (1, pl.INT64)],
                       [(0, pl.INT32), (1, pl.INT32), (3, pl.INT64)]), small=False)
    finally:
        c.destroy()


def test_context_reuse_and_result_lifetime(ctx):
    """one context, many queries (the harness reuses it for all 113), results freed out of order"""
    rng = np.random.default_rng(9)
    plans = []
    for n in (10, 1000, 50_000):
        k = rng.integers(0, n, n).astype(np.int32)
        t = pl.make_table([(pl.INT32, k), (pl.INT32, np.arange(n, dtype=np.int32))])
        plans.append(join2(t, t, True, 0, 0, [(0, pl.INT32), (1, pl.INT32)], [(0, pl.INT32), (1, pl.INT32)], [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)]))
    tabs = [[ctx.upload(t) for t in p.inputs] for p in plans]
    res = [ctx.execute_resident(p, ts) for p, ts in zip(plans, tabs)]
    for i in (1, 0, 2):
        got = res[i].to_table()
        res[i].free()
        assert pl.table_digest(got) == pl.table_digest(_oracle.execute(plans[i]))
    for ts in tabs:
        for t in ts:
            t.release()


def test_header_lying_about_non_null_count_is_decoded_from_the_bitmap(ctx):
    """A full page whose header claims n_nonnull == n_rows but whose bitmap has cleared bits: the
    reference never reads the header's non-null count for fixed-width pages
    (src/build_table.cpp:326-343), so those rows are NULL and their keys never match."""
    n = pg.rows_per_full_page(pl.INT32) * 2
    keys = np.arange(n, dtype=np.int32)
    pages = pg.pack_fixed(keys, None, pl.INT32)
    nb = (pg.rows_per_full_page(pl.INT32) + 7) // 8
    pages[0, pg.PAGE_SIZE - nb] = 0xF0  # rows 0..3 of page 0 become NULL, header still says 1984/1984
    lt = pl.ColumnarTable(n, [pl.Column(pl.INT32, pages), pl.Column(pl.INT32, pg.pack_fixed(keys, None, pl.INT32))])
    rt = pl.make_table([(pl.INT32, keys[:100].copy()), (pl.INT32, keys[:100].copy())])
    both = [(0, pl.INT32), (1, pl.INT32)]
    for resident in (False, True):
        plan = join2(lt, rt, False, 0, 0, both, both, [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)])
        want = _oracle.execute(plan)
        if resident:
            import torch

            tabs = []
            for t in plan.inputs:
                dev = [torch.from_numpy(c.pages.copy()).cuda() for c in t.columns]
                tabs.append(ctx.adopt_device(t.num_rows, [c.type for c in t.columns], [d.data_ptr() for d in dev], [d.shape[0] for d in dev], keep=dev))
            res = ctx.execute_resident(plan, tabs)
            got = res.to_table()
            res.free()
            for t in tabs:
                t.release()
        else:
            got = capi.execute(plan, ctx)
        # the values sitting in the first slots of page 0 shift: 96 rows match, not 100
        assert got.num_rows == want.num_rows
        assert pl.sorted_rows(got) == pl.sorted_rows(want)


def test_trailing_null_rows_beyond_num_rows_are_tolerated(ctx):
    """reference src/build_table.cpp:334-340: only a NON-NULL value at a row index >= num_rows
    raises "row_idx"; NULL rows past the end just advance the counter"""
    vals = np.arange(3000, dtype=np.int32)
    valid = np.ones(3000, bool)
    valid[2990:] = False
    t = pl.make_table([(pl.INT32, vals, valid), (pl.INT32, vals)])
    t.columns[1] = pl.Column(pl.INT32, pg.pack_fixed(vals[:2990], None, pl.INT32))
    t.num_rows = 2990
    other = pl.make_table([(pl.INT32, np.arange(0, 3000, 3, dtype=np.int32)), (pl.INT32, np.arange(1000, dtype=np.int32))])
    both = [(0, pl.INT32), (1, pl.INT32)]
    check(ctx, join2(t, other, True, 0, 0, both, both, [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)]), small=False)
    # ... and a non-NULL value past the end still raises, on both paths
    t.num_rows = 2980
    t.columns[1] = pl.Column(pl.INT32, pg.pack_fixed(vals[:2980], None, pl.INT32))
    plan = join2(t, other, True, 0, 0, both, both, [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)])
    with pytest.raises(capi.RjError, match="row_idx"):
        capi.execute(plan, ctx)
    with pytest.raises(RuntimeError, match="row_idx"):
        _oracle.execute(plan)


@pytest.mark.parametrize("bits,probe_dt", [(13, pl.INT32), (14, pl.INT64), (16, pl.INT64), (17, pl.INT64), (18, pl.INT32), (21, pl.INT64)])
def test_tagged_table_path_forced_bits(bits, probe_dt):
    """One key word + two-word build carry with >= 14 radix bits takes the "tagged" LDS table (13 bits: the
    generic table, kept in the sweep)
    ({19-bit tag | build index} slots + dense carries, k_join TG=1).  Duplicate build keys (the
    re-walk that emits every further match), NULL keys, a hot probe key (heavy-task path) and
    misses, against the oracle."""
    c = capi.Context(radix_bits=bits)
    try:
        rng = np.random.default_rng(500 + bits)
        nb, npr = 300_000, 700_000
        bk = rng.integers(0, 200_000, nb).astype(np.int32)  # ~1.5 copies per key
        bk[:40] = 77  # one key with 40+ copies: chains across buckets
        pk = rng.integers(0, 230_000, npr).astype(np.int32)
        pk[rng.random(npr) < 0.08] = 77
        bt = pl.make_table([(pl.INT32, bk, rng.random(nb) > 0.02), (pl.INT64, rng.integers(-(2**62), 2**62, nb).astype(np.int64))])
        pv = rng.integers(-(2**62), 2**62, npr).astype(np.int64) if probe_dt == pl.INT64 else rng.integers(-(2**31), 2**31 - 1, npr).astype(np.int32)
        pt = pl.make_table([(pl.INT32, pk, rng.random(npr) > 0.01), (probe_dt, pv)])
        for build_left in (True, False):
            if build_left:
                plan = join2(bt, pt, True, 0, 0, [(0, pl.INT32), (1, pl.INT64)], [(0, pl.INT32), (1, probe_dt)],
                             [(0, pl.INT32), (1, pl.INT64), (3, probe_dt)])
            else:
                plan = join2(pt, bt, False, 0, 0, [(0, pl.INT32), (1, probe_dt)], [(0, pl.INT32), (1, pl.INT64)],
                             [(3, pl.INT64), (1, probe_dt), (2, pl.INT32)])
            check(c, plan, small=False)
    finally:
        c.destroy()


def test_adopted_device_tables_follow_the_same_page_rules(ctx):
    """rj_table_adopt_device analyses pages on the GPU (k_page_headers / k_rows_beyond): trailing
    NULL rows beyond num_rows are tolerated, a non-NULL value beyond raises "row_idx", and a column
    whose pages are full but carry cleared bitmap bits is not addressed in place"""
    import torch

    vals = np.arange(5000, dtype=np.int32)
    valid = np.ones(5000, bool)
    valid[4990:] = False
    kpages = pg.pack_fixed(vals, valid, pl.INT32)  # NULL-bearing: irregular pages
    ppages = pg.pack_fixed(vals[:4990], None, pl.INT32)
    other = pl.make_table([(pl.INT32, np.arange(0, 5000, 5, dtype=np.int32)), (pl.INT32, np.arange(1000, dtype=np.int32))])
    both = [(0, pl.INT32), (1, pl.INT32)]

    def adopt(num_rows, pages_list):
        dev = [torch.from_numpy(p.copy()).cuda() for p in pages_list]
        return ctx.adopt_device(num_rows, [pl.INT32] * len(dev), [d.data_ptr() for d in dev], [d.shape[0] for d in dev], keep=dev)

    t = adopt(4990, [kpages, ppages])
    o = ctx.upload(other)
    host_t = pl.ColumnarTable(4990, [pl.Column(pl.INT32, kpages), pl.Column(pl.INT32, ppages)])
    plan = join2(host_t, other, True, 0, 0, both, both, [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)])
    res = ctx.execute_resident(plan, [t, o])
    got = res.to_table()
    res.free()
    assert pl.sorted_rows(got) == pl.sorted_rows(_oracle.execute(plan))
    t.release()
    with pytest.raises(capi.RjError, match="row_idx"):
        adopt(4980, [kpages, pg.pack_fixed(vals[:4980], None, pl.INT32)])  # rows 4980..4989 are non-NULL
    o.release()
