"""Differential fuzzing: random plans over random tables, GPU (through the C-ABI) vs the oracle.

Every seed builds 2-5 base tables with random schemas (INT32 / INT64 / FP64 / VARCHAR columns,
random NULL fractions, sometimes empty or zero-page tables), a random join tree over them (left
deep or bushy, random build side), join keys picked among the children's columns (usually of equal
type, occasionally of DIFFERENT types — which must give an empty result, reference
src/execute.cpp:65-71), and random output lists with reordering and duplicates.  Results are
compared as sorted row multisets."""
import os

import numpy as np
import pytest

import _oracle
from pyrj import capi
from pyrj import plan as pl

pytestmark = pytest.mark.gpu

KEYABLE = (pl.INT32, pl.INT64, pl.FP64)


@pytest.fixture(scope="module")
def ctx():
    c = capi.build_context()
    yield c
    capi.destroy_context(c)


def random_column(rng, dtype, n, key_domain, wide):
    nulls = rng.choice([0.0, 0.0, 0.0, 0.05, 0.3])
    valid = rng.random(n) >= nulls
    # half of the columns are "id"-like (unique values), the rest foreign-key-like (uniform
    # draws): keeps the many-to-many fan-out of chained joins bounded
    if n <= key_domain and rng.random() < 0.5:
        base = rng.permutation(key_domain)[:n]
    else:
        base = rng.integers(0, key_domain, n)
    if dtype == pl.INT32:
        vals = base.astype(np.int32) - (key_domain // 2 if wide else 0)  # negative keys too
        return (pl.INT32, vals, valid)
    if dtype == pl.INT64:
        vals = base.astype(np.int64) * (2**33 + 7 if wide else 1)
        return (pl.INT64, vals, valid)
    if dtype == pl.FP64:
        vals = base.astype(np.float64) / 4
        if n > 3 and rng.random() < 0.3:
            vals[:3] = [np.nan, np.inf, -np.inf]
        return (pl.FP64, vals, valid)
    strs = [None if not v else (b"s%d" % int(x)) * int(1 + x % 3) for v, x in zip(valid, rng.integers(0, key_domain, n))]
    if n > 2 and rng.random() < 0.15:
        strs[1] = b"L" * int(rng.integers(8186, 20000))  # long-string pages
    return (pl.VARCHAR, strs)


def random_table(rng, key_type, key_domain, wide):
    n = int(rng.choice([0, 1, 5, 60, 500, 1200], p=[0.02, 0.03, 0.05, 0.3, 0.4, 0.2]))
    ncols = int(rng.integers(1, 5))
    types = [key_type] + [int(rng.choice([pl.INT32, pl.INT64, pl.FP64, pl.VARCHAR])) for _ in range(ncols - 1)]
    rng.shuffle(types)
    if key_type not in types:
        types[0] = key_type
    t = pl.make_table([random_column(rng, dt, n, key_domain, wide) for dt in types])
    if n == 0 and rng.random() < 0.5:
        t = pl.ColumnarTable(0, [pl.Column(dt) for dt in types])  # typed columns, zero pages
    return t, types


def random_plan(seed, keyable=KEYABLE, key_p=(0.6, 0.25, 0.15)):
    rng = np.random.default_rng(seed)
    KEYABLE = keyable  # noqa: N806 (shadows the module default inside this plan)
    key_type = int(rng.choice(KEYABLE, p=list(key_p)))
    n_tables = int(rng.integers(2, 6))
    key_domain = int(rng.choice([300, 1500]))  # shared by the plan's tables so joins do match
    wide = bool(rng.random() < 0.3)
    p = pl.Plan()
    rels = []  # (node index, [types])
    for ti in range(n_tables):
        # occasionally a table with a different key type: joins on it must come out empty
        kt = key_type if rng.random() < 0.96 else int(rng.choice(KEYABLE))
        t, types = random_table(rng, kt, key_domain, wide)
        p.new_input(t)
        k = int(rng.integers(1, len(types) + 1))
        cols = [int(c) for c in rng.choice(len(types), size=k, replace=True)]
        if not any(types[c] == kt for c in cols):
            cols.append(types.index(kt))
        node = p.new_scan_node(ti, [(c, types[c]) for c in cols])
        rels.append((node, [types[c] for c in cols]))
    while len(rels) > 1:
        i, j = sorted(rng.choice(len(rels), size=2, replace=False))
        (ln, lt), (rn, rt) = rels[i], rels[j]

        def pick(types):
            # usually a column of the plan's key type; sometimes any keyable column, so that
            # type-mismatched and FP64/INT64 conditions are exercised as well
            pref = [c for c, dt in enumerate(types) if dt == key_type]
            anyk = [c for c, dt in enumerate(types) if dt in KEYABLE]
            return int(rng.choice(pref)) if pref and rng.random() < 0.93 else int(rng.choice(anyk))

        la, ra = pick(lt), pick(rt)
        both = lt + rt
        k = int(rng.integers(1, min(6, len(both)) + 1))
        outs = [int(c) for c in rng.choice(len(both), size=k, replace=True)]
        if not any(both[c] == key_type for c in outs):
            outs.append(la if lt[la] == key_type else (len(lt) + ra if rt[ra] == key_type else la))
        if not any(both[c] in KEYABLE for c in outs):
            outs.append(la)
        node = p.new_join_node(bool(rng.random() < 0.5), ln, rn, la, ra, [(c, both[c]) for c in outs])
        rels = [r for idx, r in enumerate(rels) if idx not in (i, j)] + [(node, [both[c] for c in outs])]
    p.root = rels[0][0]
    return p


# RJ_FUZZ_SEEDS="first:count" widens the sweep for soak runs (default: seeds 0..199)
_FIRST, _COUNT = (int(x) for x in os.environ.get("RJ_FUZZ_SEEDS", "0:200").split(":"))


@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + _COUNT))
def test_random_plan(ctx, seed):
    p = random_plan(seed)
    want = _oracle.execute(p)
    if want.num_rows > 400_000:
        pytest.skip("result too large to sort in a unit test")
    got = capi.execute(p, ctx)
    assert got.num_rows == want.num_rows
    assert [c.type for c in got.columns] == [c.type for c in want.columns]
    assert pl.canonical_rows(got) == pl.canonical_rows(want)


# The same random plans under forced radix plans: small inputs never reach the multi-pass
# partitioner on their own (their build sides take the broadcast join).  3 bits = one pass,
# 11 bits = two passes with the fine (two-digit) histogram,
# 17 bits = two passes above its LDS limit, 20 bits = three passes; nearly all partitions
# are empty or hold a single tuple.
@pytest.mark.parametrize("bits", [3, 11, 17, 20])
@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + min(_COUNT, 40)))
def test_random_plan_forced_radix(seed, bits):
    p = random_plan(seed)
    want = _oracle.execute(p)
    if want.num_rows > 400_000:
        pytest.skip("result too large to sort in a unit test")
    c = capi.Context(radix_bits=bits)
    try:
        got = capi.execute(p, c)
    finally:
        c.destroy()
    assert got.num_rows == want.num_rows
    assert pl.canonical_rows(got) == pl.canonical_rows(want)


# ... and with the XCD-aware output placement of the big passes forced on (it normally starts at
# 40 Mi tuples): per-XCD sub-ranges in the first pass, transposed grids in the later ones.
@pytest.mark.parametrize("bits", [3, 11, 17, 20])
@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + min(_COUNT, 25)))
def test_random_plan_forced_radix_xcd_placement(seed, bits):
    p = random_plan(seed)
    want = _oracle.execute(p)
    if want.num_rows > 400_000:
        pytest.skip("result too large to sort in a unit test")
    old = os.environ.get("RJ_TUNE_XCD_MIN_ROWS")
    os.environ["RJ_TUNE_XCD_MIN_ROWS"] = "0"  # read once, when the context is created
    try:
        c = capi.Context(radix_bits=bits)
    finally:
        if old is None:
            del os.environ["RJ_TUNE_XCD_MIN_ROWS"]
        else:
            os.environ["RJ_TUNE_XCD_MIN_ROWS"] = old
    try:
        got = capi.execute(p, c)
    finally:
        c.destroy()
    assert got.num_rows == want.num_rows
    assert pl.canonical_rows(got) == pl.canonical_rows(want)


# Payload columns travel with the key when they fit the carry words (CARRY_WIDE, the default since
# round 3); with RJ_TUNE_WIDE_CARRY=0 a row index travels instead and every column is gathered
# afterwards (k_gather) — that path stays covered by the same plans.
@pytest.mark.parametrize("bits", [0, 11])
@pytest.mark.parametrize("seed", range(_FIRST, _FIRST + min(_COUNT, 60)))
def test_random_plan_row_index_carries(seed, bits):
    p = random_plan(seed)
    want = _oracle.execute(p)
    if want.num_rows > 400_000:
        pytest.skip("result too large to sort in a unit test")
    os.environ["RJ_TUNE_WIDE_CARRY"] = "0"  # read once, when the context is created
    try:
        c = capi.Context(radix_bits=bits)
    finally:
        del os.environ["RJ_TUNE_WIDE_CARRY"]
    try:
        got = capi.execute(p, c)
    finally:
        c.destroy()
    assert got.num_rows == want.num_rows
    assert pl.canonical_rows(got) == pl.canonical_rows(want)


# VARCHAR join keys (reference hash_join_omp<std::string>) in the mix: the same generator with
# VARCHAR as the plan's key type most of the time; own seed range so that the plans above stay
# what they were.  Also under a forced multi-pass radix plan (the 64-bit string hashes then go
# through the partitioner instead of the broadcast join).
@pytest.mark.parametrize("seed", range(50_000, 50_060))
def test_random_plan_with_varchar_keys(ctx, seed):
    p = random_plan(seed, keyable=KEYABLE + (pl.VARCHAR,), key_p=(0.15, 0.1, 0.05, 0.7))
    want = _oracle.execute(p)
    if want.num_rows > 400_000:
        pytest.skip("result too large to sort in a unit test")
    got = capi.execute(p, ctx)
    assert got.num_rows == want.num_rows
    assert [c.type for c in got.columns] == [c.type for c in want.columns]
    assert pl.canonical_rows(got) == pl.canonical_rows(want)


@pytest.mark.parametrize("seed", range(50_000, 50_020))
def test_random_plan_with_varchar_keys_forced_radix(seed):
    p = random_plan(seed, keyable=KEYABLE + (pl.VARCHAR,), key_p=(0.15, 0.1, 0.05, 0.7))
    want = _oracle.execute(p)
    if want.num_rows > 400_000:
        pytest.skip("result too large to sort in a unit test")
    c = capi.Context(radix_bits=11)
    try:
        got = capi.execute(p, c)
    finally:
        c.destroy()
    assert got.num_rows == want.num_rows
    assert pl.canonical_rows(got) == pl.canonical_rows(want)
