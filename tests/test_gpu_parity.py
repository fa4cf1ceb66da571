"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle and the
reference's golden unit cases.  Results are multisets of rows (the reference's
row order is nondeterministic, src/execute.cpp:252-261), so rows are compared
sorted (small) or through an order-independent digest (large).  Integer / byte
work: the bar is bit-exact."""
import numpy as np
import pytest

import _oracle
from _golden import build_plan, expected_rows, load_cases
from pyrj import capi
from pyrj import plan as pl

pytestmark = pytest.mark.gpu

CASES = load_cases()


@pytest.fixture(scope="module")
def ctx():
    c = capi.build_context()
    yield c
    capi.destroy_context(c)


def check_vs_oracle(ctx, plan, small=True):
    got = capi.execute(plan, ctx)
    want = _oracle.execute(plan)
    assert got.num_rows == want.num_rows
    assert [c.type for c in got.columns] == [c.type for c in want.columns]
    if small:
        assert pl.sorted_rows(got) == pl.sorted_rows(want)
    else:
        assert pl.table_digest(got) == pl.table_digest(want)
    return got


# ------------------------------------------------------------ golden vectors --
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_reference_unit_cases(ctx, case):
    """reference tests/unit_tests.cpp, all 8 TEST_CASEs"""
    plan = build_plan(case)
    res = capi.execute(plan, ctx)
    exp = case["expect"]
    assert res.num_rows == exp["num_rows"]
    assert [c.type for c in res.columns] == [pl.TYPE_IDS[t] for t in exp["types"]]
    if exp["num_rows"] == 0:
        assert all(c.pages.shape[0] == 0 for c in res.columns)
    assert pl.sorted_rows(res) == expected_rows(case)


# ------------------------------------------------------------- single joins --
def two_table_plan(lt, rt, build_left, left_attr, right_attr, l_out, r_out, outs):
    p = pl.Plan()
    p.new_scan_node(0, l_out)
    p.new_scan_node(1, r_out)
    p.new_join_node(build_left, 0, 1, left_attr, right_attr, outs)
    p.new_input(lt)
    p.new_input(rt)
    p.root = 2
    return p


@pytest.mark.parametrize("build_left", [True, False])
@pytest.mark.parametrize("nb,npr,null_frac", [(1000, 1000, 0.0), (3000, 7000, 0.1), (50_000, 200_000, 0.0), (70_000, 30_000, 0.3)])
def test_int32_join_payloads(ctx, build_left, nb, npr, null_frac):
    rng = np.random.default_rng(nb + npr)
    bk = rng.integers(-(2**31), 2**31 - 1, nb).astype(np.int32)
    pk = np.concatenate([rng.choice(bk, npr // 2), rng.integers(-(2**31), 2**31 - 1, npr - npr // 2).astype(np.int32)])
    rng.shuffle(pk)
    bv = rng.random(nb) >= null_frac
    pv = rng.random(npr) >= null_frac
    bt = pl.make_table([(pl.INT32, bk, bv), (pl.INT32, np.arange(nb, dtype=np.int32))])
    pt = pl.make_table([(pl.INT32, pk, pv), (pl.INT32, -np.arange(npr, dtype=np.int32))])
    both = [(0, pl.INT32), (1, pl.INT32)]
    if build_left:
        plan = two_table_plan(bt, pt, True, 0, 0, both, both, [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)])
    else:
        plan = two_table_plan(pt, bt, False, 0, 0, both, both, [(2, pl.INT32), (3, pl.INT32), (1, pl.INT32), (0, pl.INT32)])
    check_vs_oracle(ctx, plan, small=nb + npr <= 20_000)


def test_duplicates_multiply_and_build_chunks(ctx):
    """dup x dup; one build key with 10k copies forces several LDS table chunks"""
    rng = np.random.default_rng(11)
    bk = np.concatenate([np.full(10_000, 42, np.int32), rng.integers(0, 500, 3000).astype(np.int32)])
    pk = np.concatenate([np.full(7, 42, np.int32), rng.integers(0, 600, 2000).astype(np.int32)])
    bt = pl.make_table([(pl.INT32, bk), (pl.INT32, np.arange(bk.size, dtype=np.int32))])
    pt = pl.make_table([(pl.INT32, pk), (pl.INT32, np.arange(pk.size, dtype=np.int32))])
    both = [(0, pl.INT32), (1, pl.INT32)]
    plan = two_table_plan(bt, pt, True, 0, 0, both, both, [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)])
    got = check_vs_oracle(ctx, plan, small=False)
    assert got.num_rows > 70_000


def test_probe_skew_heavy_partitions(ctx):
    """Zipf-like probe side: one key owns 300k probe tuples (> JN_HEAVY, task splitting)"""
    rng = np.random.default_rng(13)
    nb = 20_000
    bk = rng.permutation(nb).astype(np.int32)
    pk = np.concatenate([np.full(300_000, 7, np.int32), np.full(50_000, 9, np.int32), rng.integers(0, nb, 100_000).astype(np.int32)])
    rng.shuffle(pk)
    bt = pl.make_table([(pl.INT32, bk), (pl.INT32, np.arange(nb, dtype=np.int32))])
    pt = pl.make_table([(pl.INT32, pk), (pl.INT32, np.arange(pk.size, dtype=np.int32))])
    both = [(0, pl.INT32), (1, pl.INT32)]
    plan = two_table_plan(bt, pt, True, 0, 0, both, both, [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)])
    got = check_vs_oracle(ctx, plan, small=False)
    assert got.num_rows == pk.size


def test_output_larger_than_inputs_retries(ctx):
    """many-to-many: |out| >> max(|L|,|R|) exercises the exact-size second probe run"""
    bk = np.repeat(np.arange(50, dtype=np.int32), 40)
    pk = np.repeat(np.arange(50, dtype=np.int32), 60)
    bt = pl.make_table([(pl.INT32, bk), (pl.INT32, np.arange(bk.size, dtype=np.int32))])
    pt = pl.make_table([(pl.INT32, pk), (pl.INT32, np.arange(pk.size, dtype=np.int32))])
    both = [(0, pl.INT32), (1, pl.INT32)]
    plan = two_table_plan(bt, pt, True, 0, 0, both, both, [(1, pl.INT32), (3, pl.INT32), (0, pl.INT32)])
    got = check_vs_oracle(ctx, plan, small=False)
    assert got.num_rows == 50 * 40 * 60


@pytest.mark.parametrize("ktype", [pl.INT64, pl.FP64])
def test_wide_keys_and_int64_payload(ctx, ktype):
    rng = np.random.default_rng(17 + ktype)
    nb, npr = 20_000, 50_000
    if ktype == pl.INT64:
        bk = rng.integers(-(2**62), 2**62, nb).astype(np.int64)
        pk = np.concatenate([rng.choice(bk, npr // 2), rng.integers(-(2**62), 2**62, npr - npr // 2).astype(np.int64)])
    else:
        bk = np.round(rng.standard_normal(nb) * 1000) / 8
        bk[:4] = [0.0, -0.0, np.nan, np.inf]
        pk = np.concatenate([rng.choice(bk, npr // 2), np.round(rng.standard_normal(npr - npr // 2) * 1000) / 8])
        pk[:4] = [-0.0, 0.0, np.nan, np.inf]
    bv = rng.random(nb) >= 0.05
    bt = pl.make_table([(ktype, bk, bv), (pl.INT64, rng.integers(-(2**60), 2**60, nb).astype(np.int64))])
    pt = pl.make_table([(pl.FP64, rng.standard_normal(npr)), (ktype, pk)])
    plan = two_table_plan(
        bt, pt, True, 0, 1, [(0, ktype), (1, pl.INT64)], [(0, pl.FP64), (1, ktype)],
        [(1, pl.INT64), (2, pl.FP64), (3, ktype), (0, ktype)],
    )
    check_vs_oracle(ctx, plan, small=False)


def test_key_type_mismatch_is_empty(ctx):
    bt = pl.table_from_rows([(1,), (2,)], [pl.INT64])
    pt = pl.table_from_rows([(1,), (2,)], [pl.INT32])
    plan = two_table_plan(bt, pt, True, 0, 0, [(0, pl.INT64)], [(0, pl.INT32)], [(0, pl.INT64), (1, pl.INT32)])
    got = check_vs_oracle(ctx, plan)
    assert got.num_rows == 0


def test_nullable_payloads_and_column_reorder(ctx):
    rng = np.random.default_rng(23)
    nb, npr = 5000, 9000
    bk = rng.integers(0, 4000, nb).astype(np.int32)
    pk = rng.integers(0, 4500, npr).astype(np.int32)
    bt = pl.make_table([
        (pl.INT32, rng.integers(0, 99, nb).astype(np.int32), rng.random(nb) > 0.4),
        (pl.INT32, bk, rng.random(nb) > 0.1),
        (pl.INT64, rng.integers(0, 2**40, nb).astype(np.int64), rng.random(nb) > 0.5),
    ])
    pt = pl.make_table([(pl.INT32, pk), (pl.FP64, rng.standard_normal(npr), rng.random(npr) > 0.2)])
    plan = two_table_plan(
        bt, pt, False, 1, 0,
        [(2, pl.INT64), (1, pl.INT32), (0, pl.INT32)], [(0, pl.INT32), (1, pl.FP64)],
        [(4, pl.FP64), (0, pl.INT64), (2, pl.INT32), (1, pl.INT32), (3, pl.INT32), (0, pl.INT64)],
    )
    check_vs_oracle(ctx, plan, small=False)


def test_varchar_payload_and_multi_join_tree(ctx):
    rng = np.random.default_rng(29)
    a = [(int(k), int(k) * 10 if k % 7 else None) for k in rng.integers(0, 300, 2000)]
    b = [(int(k), (f"name-{int(k)}-{i}" if i % 11 else None)) for i, k in enumerate(rng.integers(0, 300, 1500))]
    b[5] = (b[5][0], "L" * 9000)  # long string pages
    c = [(int(k) if k % 13 else None,) for k in rng.integers(0, 300, 800)]
    p = pl.Plan()
    sa = p.new_scan_node(0, [(0, pl.INT32), (1, pl.INT32)])
    sb = p.new_scan_node(1, [(1, pl.VARCHAR), (0, pl.INT32)])
    sc = p.new_scan_node(2, [(0, pl.INT32)])
    j1 = p.new_join_node(False, sa, sb, 0, 1, [(1, pl.INT32), (2, pl.VARCHAR), (0, pl.INT32), (0, pl.INT32)])
    j2 = p.new_join_node(True, sc, j1, 0, 2, [(2, pl.VARCHAR), (0, pl.INT32), (1, pl.INT32), (4, pl.INT32), (2, pl.VARCHAR)])
    p.new_input(pl.table_from_rows(a, [pl.INT32, pl.INT32]))
    p.new_input(pl.table_from_rows(b, [pl.INT32, pl.VARCHAR]))
    p.new_input(pl.table_from_rows(c, [pl.INT32]))
    p.root = j2
    check_vs_oracle(ctx, p)


def test_root_scan_passthrough(ctx):
    t = pl.table_from_rows([(1, "a", 2.5), (None, None, None), (3, "ccc", -1.0)], [pl.INT32, pl.VARCHAR, pl.FP64])
    p = pl.Plan()
    p.new_scan_node(0, [(2, pl.FP64), (0, pl.INT32), (1, pl.VARCHAR), (0, pl.INT32)])
    p.new_input(t)
    p.root = 0
    check_vs_oracle(ctx, p)


def test_mid_size_two_pass_vs_oracle(ctx):
    """2M ⋈ 4M: two radix passes (bits > 9); oracle finishes in a few seconds"""
    rng = np.random.default_rng(31)
    nb, npr = 2_000_000, 4_000_000
    bk = rng.permutation(nb).astype(np.int32)
    pk = rng.integers(0, nb, npr).astype(np.int32)
    bt = pl.make_table([(pl.INT32, bk), (pl.INT32, np.arange(nb, dtype=np.int32))])
    pt = pl.make_table([(pl.INT32, pk), (pl.INT32, np.arange(npr, dtype=np.int32))])
    both = [(0, pl.INT32), (1, pl.INT32)]
    plan = two_table_plan(bt, pt, True, 0, 0, both, both, [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)])
    got = check_vs_oracle(ctx, plan, small=False)
    assert got.num_rows == npr


# -------------------------------------------------------------------- errors --
def test_varchar_key_tiny(ctx):
    """VARCHAR join keys (reference hash_join_omp<std::string>, src/execute.cpp:278): see
    tests/test_gpu_varchar_keys.py for the real coverage"""
    t = pl.table_from_rows([("a",), ("b",), (None,), ("a",)], [pl.VARCHAR])
    plan = two_table_plan(t, t, True, 0, 0, [(0, pl.VARCHAR)], [(0, pl.VARCHAR)], [(0, pl.VARCHAR)])
    got = capi.execute(plan, ctx)
    assert pl.sorted_rows(got) == pl.sorted_rows(_oracle.execute(plan))
    assert got.num_rows == 5  # a x a: 4, b x b: 1, NULL never matches


def test_pages_with_more_rows_than_declared_raise_row_idx(ctx):
    """reference: throw std::runtime_error("row_idx") (build_table.cpp:334-336)"""
    t = pl.make_table([(pl.INT32, np.arange(10, dtype=np.int32))])
    t.num_rows = 5
    plan = two_table_plan(t, t, True, 0, 0, [(0, pl.INT32)], [(0, pl.INT32)], [(0, pl.INT32)])
    with pytest.raises(capi.RjError, match="row_idx"):
        capi.execute(plan, ctx)
    with pytest.raises(RuntimeError, match="row_idx"):
        _oracle.execute(plan)


# ------------------------------------------------- resident / sharded entry points
def test_resident_tables_and_virtual_ranks(ctx):
    """Inputs adopted from HBM; then the 2-rank sharded path run as two virtual ranks on one
    GPU: stage A per shard, exchange by pointer arithmetic, stage B per rank."""
    import torch

    rng = np.random.default_rng(37)
    nb, npr = 300_000, 500_000
    bk = rng.permutation(nb).astype(np.int32)
    pk = rng.integers(0, nb, npr).astype(np.int32)
    bt = pl.make_table([(pl.INT32, bk), (pl.INT32, np.arange(nb, dtype=np.int32))])
    pt = pl.make_table([(pl.INT32, pk), (pl.INT32, np.arange(npr, dtype=np.int32))])
    both = [(0, pl.INT32), (1, pl.INT32)]
    plan = two_table_plan(bt, pt, True, 0, 0, both, both, [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)])
    want = _oracle.execute(plan)

    def adopt(t):
        devs = [torch.from_numpy(c.pages).cuda() for c in t.columns]
        torch.cuda.synchronize()
        return ctx.adopt_device(t.num_rows, [c.type for c in t.columns], [d.data_ptr() for d in devs], [c.pages.shape[0] for c in t.columns], keep=devs)

    tb, tp = adopt(bt), adopt(pt)
    res = ctx.execute_resident(plan, [tb, tp])
    assert res.device_pages(0) is not None
    got = res.to_table()
    assert pl.table_digest(got) == pl.table_digest(want)
    res.free()

    # virtual 2-rank run: shard both relations in halves (page granular)
    from pyrj import dist

    parts = dist.virtual_rank_join(ctx, bt, pt, n_ranks=2)
    rows = sum(p.num_rows for p in parts)
    assert rows == want.num_rows
    # the digest is a (count, sum, xor) of row hashes, so per-rank digests combine
    digs = [pl.table_digest(p) for p in parts]
    n = sum(d[0] for d in digs)
    s = sum(d[1] for d in digs) % (1 << 64)
    x = 0
    for d in digs:
        x ^= d[2]
    assert (n, s, x) == pl.table_digest(want)
    tb.release()
    tp.release()


def test_stage_a_matches_host_hash_restatement(ctx):
    """rj_shard_partition: tuples grouped by owner rank = top bits of the library's hash
    (pyrj.hashing restates it in numpy); keys come out hashed and un-hash to the input."""
    from pyrj import dist, hashing

    rng = np.random.default_rng(41)
    n = 200_000
    k = rng.integers(-(2**31), 2**31 - 1, n).astype(np.int32)
    v = rng.random(n) > 0.1
    t = pl.make_table([(pl.INT32, k, v), (pl.INT32, np.arange(n, dtype=np.int32))])
    tb = ctx.upload(t)
    ops = dist.GpuOps(ctx)
    for n_ranks in (1, 2, 8):
        hk, carry, counts = ops.partition(tb, n, n_ranks)
        hk, carry = hk.cpu().numpy(), carry.cpu().numpy()
        assert sum(counts) == int(v.sum()) == hk.shape[0]
        keys = hashing.unfmix32(hk.view(np.uint32)).view(np.int32)
        assert np.array_equal(keys, k[carry])  # payload = original row index
        assert np.array_equal(np.sort(carry), np.nonzero(v)[0])
        owner = hashing.owner_rank(keys, n_ranks)
        assert np.array_equal(owner, np.repeat(np.arange(n_ranks), counts))
    tb.release()
