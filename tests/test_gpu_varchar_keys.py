"""VARCHAR join keys (reference hash_join_omp<std::string>, src/execute.cpp:33-38,278): the key
strings are hashed on the device (FNV-1a 64), joined as 64-bit keys, and every joined pair is
compared byte for byte, so hash collisions cannot add rows.  Against the oracle, row by row."""
import os

import numpy as np
import pytest

import _oracle
from pyrj import capi
from pyrj import plan as pl

pytestmark = pytest.mark.gpu


def ctx_with(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return capi.build_context()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def words(rng, n, vocab, null_frac=0.05, long_every=0):
    out = []
    for i in range(n):
        if rng.random() < null_frac:
            out.append(None)
        elif long_every and i % long_every == 0:
            k = int(rng.integers(0, 4))
            out.append((b"L%d-" % k) + b"x" * (9000 + 100 * k))  # long-string page chains, few distinct values
        else:
            out.append(b"w%d" % int(rng.integers(0, vocab)) if rng.random() > 0.02 else b"")
    return out


def check(ctx, plan):
    got = capi.execute(plan, ctx)
    want = _oracle.execute(plan)
    assert got.num_rows == want.num_rows
    assert [c.type for c in got.columns] == [c.type for c in want.columns]
    assert pl.sorted_rows(got) == pl.sorted_rows(want)
    return got


def two(bt, pt, build_left, bcols, pcols, outs, bkey=0, pkey=0):
    p = pl.Plan()
    if build_left:
        p.new_scan_node(0, bcols)
        p.new_scan_node(1, pcols)
        p.new_join_node(True, 0, 1, bkey, pkey, outs)
        p.new_input(bt)
        p.new_input(pt)
    else:
        p.new_scan_node(0, pcols)
        p.new_scan_node(1, bcols)
        p.new_join_node(False, 0, 1, pkey, bkey, outs)
        p.new_input(pt)
        p.new_input(bt)
    p.root = 2
    return p


@pytest.mark.parametrize("build_left", [True, False])
@pytest.mark.parametrize("env", [{}, {"RJ_DEBUG_VKEY_HASH_BITS": "6"}], ids=["full-hash", "6-bit-hash-collisions"])
def test_varchar_keys_small_build(build_left, env):
    """broadcast-join sized build side; with a 6-bit hash nearly every pair the kernel joins is a
    collision that the byte-for-byte check has to throw out again (verify + compact path)"""
    rng = np.random.default_rng(60)
    nb, npr = 900, 20_000
    bt = pl.make_table([(pl.VARCHAR, words(rng, nb, 400, long_every=97)), (pl.INT32, np.arange(nb, dtype=np.int32))])
    pt = pl.make_table([(pl.INT64, rng.integers(0, 2**40, npr).astype(np.int64)), (pl.VARCHAR, words(rng, npr, 450, long_every=501))])
    c = ctx_with(env)
    try:
        if build_left:
            outs = [(0, pl.VARCHAR), (1, pl.INT32), (2, pl.INT64), (3, pl.VARCHAR)]
        else:
            outs = [(1, pl.VARCHAR), (0, pl.INT64), (3, pl.INT32)]
        check(c, two(bt, pt, build_left, [(0, pl.VARCHAR), (1, pl.INT32)], [(0, pl.INT64), (1, pl.VARCHAR)], outs, 0, 1))
    finally:
        capi.destroy_context(c)


@pytest.mark.parametrize("env", [{}, {"RJ_DEBUG_VKEY_HASH_BITS": "10"}], ids=["full-hash", "10-bit-hash-collisions"])
def test_varchar_keys_partitioned(env):
    """build side above the broadcast limit: radix passes over the 64-bit string hashes"""
    rng = np.random.default_rng(61)
    nb, npr = 60_000, 150_000
    bt = pl.make_table([(pl.VARCHAR, words(rng, nb, 50_000)), (pl.VARCHAR, [b"p%d" % i for i in range(nb)])])
    pt = pl.make_table([(pl.VARCHAR, words(rng, npr, 55_000)), (pl.INT32, np.arange(npr, dtype=np.int32))])
    c = ctx_with(env)
    try:
        check(c, two(bt, pt, True, [(0, pl.VARCHAR), (1, pl.VARCHAR)], [(0, pl.VARCHAR), (1, pl.INT32)],
                     [(1, pl.VARCHAR), (3, pl.INT32), (0, pl.VARCHAR)]))
    finally:
        capi.destroy_context(c)


def test_varchar_key_of_an_intermediate_result_and_type_mismatch():
    """(A ⋈ B on INT32) ⋈ C on a VARCHAR column that came out of the first join; and a VARCHAR
    build key against an INT32 probe key matches nothing (src/execute.cpp:65-71)"""
    rng = np.random.default_rng(62)
    a = pl.make_table([(pl.INT32, rng.permutation(5000).astype(np.int32)), (pl.VARCHAR, words(rng, 5000, 300))])
    b = pl.make_table([(pl.INT32, rng.integers(0, 5000, 12_000).astype(np.int32))])
    cdim = pl.make_table([(pl.VARCHAR, [b"w%d" % i for i in range(0, 300, 2)] + [None, b""]), (pl.INT64, np.arange(152, dtype=np.int64))])
    p = pl.Plan()
    sa = p.new_scan_node(0, [(0, pl.INT32), (1, pl.VARCHAR)])
    sb = p.new_scan_node(1, [(0, pl.INT32)])
    j1 = p.new_join_node(True, sa, sb, 0, 0, [(1, pl.VARCHAR), (0, pl.INT32)])
    sc = p.new_scan_node(2, [(0, pl.VARCHAR), (1, pl.INT64)])
    j2 = p.new_join_node(False, j1, sc, 0, 0, [(0, pl.VARCHAR), (1, pl.INT32), (3, pl.INT64)])
    for t in (a, b, cdim):
        p.new_input(t)
    p.root = j2
    c = capi.build_context()
    try:
        check(c, p)
        mism = two(cdim, b, True, [(0, pl.VARCHAR), (1, pl.INT64)], [(0, pl.INT32)], [(1, pl.INT64), (2, pl.INT32)])
        got = check(c, mism)
        assert got.num_rows == 0
    finally:
        capi.destroy_context(c)
