"""VARCHAR gather + page encode on the device (csrc/rj_varchar_dev.hip): root
VARCHAR columns of large results are materialised on the GPU.  RJ_TUNE_VARCHAR_DEV=1 forces that
path for every VARCHAR result column, so the small cases below exercise it too; everything is
compared with the oracle row by row (the page layout may differ from the oracle's — chunks start
fresh pages — but decodes to the same rows)."""
import numpy as np
import pytest

import _oracle
from pyrj import capi, job
from pyrj import pages as pg
from pyrj import plan as pl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import os

    old = os.environ.get("RJ_TUNE_VARCHAR_DEV")
    os.environ["RJ_TUNE_VARCHAR_DEV"] = "1"  # read once, when the context is created
    c = capi.build_context()
    if old is None:
        del os.environ["RJ_TUNE_VARCHAR_DEV"]
    else:
        os.environ["RJ_TUNE_VARCHAR_DEV"] = old
    yield c
    capi.destroy_context(c)


def strings(rng, n, null_frac=0.1, long_every=0, max_len=60):
    out = []
    for i in range(n):
        r = rng.random()
        if r < null_frac:
            out.append(None)
        elif long_every and i % long_every == long_every - 1:
            ln = int(rng.choice([8185, 8186, 8188, 8189, 9000, 16376, 16377, 30000]))
            out.append((b"%d:" % i) + bytes(rng.integers(97, 123, ln - len(b"%d:" % i), dtype=np.uint8)))
        elif r < null_frac + 0.05:
            out.append(b"")
        else:
            out.append((b"s%d-" % i) + bytes(rng.integers(65, 91, int(rng.integers(0, max_len)), dtype=np.uint8)))
    return out


def plan_with_varchar(bt, pt, bcols, pcols, outs):
    p = pl.Plan()
    p.new_scan_node(0, bcols)
    p.new_scan_node(1, pcols)
    p.new_join_node(True, 0, 1, 0, 0, outs)
    p.new_input(bt)
    p.new_input(pt)
    p.root = 2
    return p


def check(ctx, plan):
    got = capi.execute(plan, ctx)
    want = _oracle.execute(plan)
    assert got.num_rows == want.num_rows
    assert [c.type for c in got.columns] == [c.type for c in want.columns]
    assert pl.sorted_rows(got) == pl.sorted_rows(want)
    return got


def test_varchar_payloads_nulls_empty_and_long_strings(ctx):
    rng = np.random.default_rng(41)
    nb, npr = 6000, 20000
    bt = pl.make_table([(pl.INT32, rng.permutation(nb).astype(np.int32)), (pl.VARCHAR, strings(rng, nb, long_every=97))])
    pt = pl.make_table([(pl.INT32, rng.integers(0, nb + 500, npr).astype(np.int32)), (pl.VARCHAR, strings(rng, npr, null_frac=0.3))])
    got = check(ctx, plan_with_varchar(bt, pt, [(0, pl.INT32), (1, pl.VARCHAR)], [(0, pl.INT32), (1, pl.VARCHAR)],
                                       [(1, pl.VARCHAR), (0, pl.INT32), (3, pl.VARCHAR)]))
    # long strings came out as 0xffff / 0xfffe page chains
    hdr = got.columns[0].pages[:, :2].copy().view(np.uint16)[:, 0]
    assert (hdr == 0xFFFF).any() and (hdr == 0xFFFE).any()


def test_all_null_and_all_empty_columns(ctx):
    rng = np.random.default_rng(42)
    n = 70_000  # all-NULL pages hold up to 65k rows each
    bt = pl.make_table([(pl.INT32, np.arange(n, dtype=np.int32)), (pl.VARCHAR, [None] * n), (pl.VARCHAR, [b""] * n)])
    pt = pl.make_table([(pl.INT32, rng.integers(0, n, n).astype(np.int32))])
    check(ctx, plan_with_varchar(bt, pt, [(0, pl.INT32), (1, pl.VARCHAR), (2, pl.VARCHAR)], [(0, pl.INT32)],
                                 [(1, pl.VARCHAR), (2, pl.VARCHAR), (3, pl.INT32)]))


def test_half_million_row_varchar_result_default_threshold():
    """above the default threshold (200 k rows) a plain context takes the device path by itself"""
    rng = np.random.default_rng(43)
    nb, npr = 300_000, 520_000
    titles = strings(rng, nb, null_frac=0.05, long_every=50_000, max_len=40)
    bt = pl.make_table([(pl.INT32, rng.permutation(nb).astype(np.int32)), (pl.VARCHAR, titles)])
    pt = pl.make_table([(pl.INT32, rng.integers(0, nb, npr).astype(np.int32)), (pl.INT32, np.arange(npr, dtype=np.int32))])
    c = capi.build_context()
    try:
        check(c, plan_with_varchar(bt, pt, [(0, pl.INT32), (1, pl.VARCHAR)], [(0, pl.INT32), (1, pl.INT32)],
                                   [(1, pl.VARCHAR), (3, pl.INT32)]))
    finally:
        capi.destroy_context(c)


@pytest.mark.parametrize("name", ["1a", "10c", "13d", "16b", "17a", "22c", "33c", "6f"])
def test_job_plans_with_device_varchar(ctx, name):
    fx = job.load_fixture()
    tables = job.make_tables(fx["schema"], seed=5, scale=3.0)
    check(ctx, job.build_plan(fx["queries"][name], fx["schema"], tables))
