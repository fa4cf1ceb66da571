"""The benchmarked configurations at FULL size, device-resident, checked in closed form
(pyrj/workloads.py::verify_pk_fk: every probe row exactly once, with its own key and payload,
next to the build row that holds that key) — plus one HIP-vs-oracle digest at 50 M x 50 M rows
for the path that carries two-word (INT64) payloads next to a one-word key, whose partitions are
12-byte tuples."""
import numpy as np
import pytest

import _oracle
from pyrj import capi
from pyrj import plan as pl
from pyrj import workloads as wl

pytestmark = pytest.mark.gpu


def run_and_verify(name):
    import torch

    ctx = capi.build_context()
    try:
        hbm = ctx.device_info()["hbm_bytes"]
        rows = wl.WORKLOADS[name]["rows"]
        if hbm < rows * 160:  # inputs + two ping-pong partition buffers per side + result
            pytest.skip(f"{name} needs more HBM than this device has")
        rel = wl.make_relations(name, torch.device("cuda"))
        R = wl.adopt(ctx, [rel.rk, rel.rp])
        S = wl.adopt(ctx, [rel.sk, rel.sp])
        res = ctx.execute_resident(wl.join_plan(rel.payload_type), [R, S])
        digest = wl.verify_pk_fk(res, rel)
        assert digest["rows"] == rows
        res.free()
        R.release()
        S.release()
    finally:
        capi.destroy_context(ctx)
        torch.cuda.empty_cache()


@pytest.mark.parametrize("side_arrays", [False, True])
@pytest.mark.parametrize("name", ["config2", "config3"])
def test_between_the_benchmarked_sizes_150m(name, side_arrays):
    """150 M ⋈ 150 M of both benchmark shapes: 2^16 partitions — two plain-histogram passes of
    2^8 (above the fine histogram's limit, below the 2^9-way passes of the 1 B runs), with the
    XCD-aware output placement on (>= 40 Mi tuples) and, for the 12-byte tuples, the tagged join
    table at 16 bits.  side_arrays: the optional digit side arrays between the passes
    (RJ_TUNE_PACKED_SIDE / RJ_TUNE_AOS_MID = 1; net-neutral, off by default) — the second pass'
    histogram then reads 16-bit digits.  Closed-form verification as for the full sizes."""
    import os

    import torch

    env = {"RJ_TUNE_PACKED_SIDE": "1", "RJ_TUNE_AOS_MID": "1"} if side_arrays else {}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)  # read once, when the context is created
    try:
        ctx = capi.build_context()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    try:
        rel = wl.make_relations(name, torch.device("cuda"), rows=150_000_000)
        R = wl.adopt(ctx, [rel.rk, rel.rp])
        S = wl.adopt(ctx, [rel.sk, rel.sp])
        res = ctx.execute_resident(wl.join_plan(rel.payload_type), [R, S])
        digest = wl.verify_pk_fk(res, rel)
        assert digest["rows"] == 150_000_000
        res.free()
        R.release()
        S.release()
    finally:
        capi.destroy_context(ctx)
        torch.cuda.empty_cache()


def test_config2_100m_uniform_full_size():
    """BASELINE config 2: 100 M ⋈ 100 M INT32 uniform keys, INT32 payloads (fine histogram,
    packed {key, carry} pairs, two 2^7/2^8-way passes)"""
    run_and_verify("config2")


def test_config3_1b_zipf_int64_full_size():
    """BASELINE config 3: 1 B ⋈ 1 B, Zipf-0.9 probe keys (the hottest key owns ~1.3 % of the
    probe side: heavy-task path), INT64 payloads (two-word carries), two 2^9-way passes"""
    run_and_verify("config3")


def test_uniform_1b_full_size():
    """1 B ⋈ 1 B INT32 uniform keys, INT32 payloads (2^18 partitions of packed 8-byte pairs: two 2^9-way
    passes with plain histograms, the pairs between them in blocks of 256 keys + 256 carries —
    BlockedLoader — whose key halves are all the second histogram reads)"""
    run_and_verify("uniform1b")


def test_two_word_carry_path_50m_vs_oracle():
    """KW = 1 / CW = 2 (INT32 key + INT64 payload on both sides) at 50 M x 50 M with a Zipf-0.9
    probe side: HIP through the C-ABI (host pages in and out) against the oracle, by the
    order-independent digest of the result rows"""
    rng = np.random.default_rng(77)
    n = 50_000_000
    bk = rng.permutation(n).astype(np.int32)
    w = 1.0 / np.arange(1, n + 1, dtype=np.float64) ** 0.9
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    ranks = np.searchsorted(cdf, rng.random(n), side="right").clip(max=n - 1).astype(np.int64)
    del w, cdf
    pk = ((ranks * 7919 + 13) % n).astype(np.int32)
    del ranks
    bt = pl.make_table([(pl.INT32, bk), (pl.INT64, rng.integers(-(2**62), 2**62, n).astype(np.int64))])
    pt = pl.make_table([(pl.INT32, pk), (pl.INT64, rng.integers(-(2**62), 2**62, n).astype(np.int64))])
    p = wl.join_plan(pl.INT64)
    p.new_input(bt)
    p.new_input(pt)
    ctx = capi.build_context()
    try:
        got = capi.execute(p, ctx)
    finally:
        capi.destroy_context(ctx)
    dg = pl.table_digest(got)
    del got
    want = _oracle.execute(p)
    assert want.num_rows == n
    assert dg == pl.table_digest(want)


def test_sharded_config3_shape_eight_virtual_ranks_200m():
    """The sharded path at a size where every count is far beyond 16 bits: 200 M ⋈ 200 M of
    config 3's shape (Zipf-0.9 probe keys, INT64 payloads) over eight virtual ranks on this one
    GPU (rj_execute_sharded, peer-copy transport).  Closed form: every rank's output rows name a
    build row that holds their key (the build keys are a bijection of the global row index),
    and over all ranks the result has |S| rows with the probe side's key sum and payload checksum."""
    import torch

    world, total = 8, 200_000_000
    ctx = capi.Context(devices=[0] * world)
    tables, rels = [], []
    try:
        for r in range(world):
            rel = wl.make_relations("config3", torch.device("cuda"), rows=total, rank=r, world=world, sharded=True)
            lane = ctx.lane(r)
            tables.append([wl.adopt(lane, [rel.rk, rel.rp]), wl.adopt(lane, [rel.sk, rel.sp])])
            rels.append(rel)
        results = ctx.execute_sharded(wl.join_plan(pl.INT64), tables)
        rows = 0
        sum_key = want_key = 0
        sum_mix = want_mix = 0
        mask = 0xFFFFFFFFFFFFFFFF
        for res, rel in zip(results, rels):
            n_out = res.num_rows
            rows += n_out
            assert n_out > 0.5 * total / world  # hash sharding spreads even a Zipf probe side
            key2d, _ = wl.result_column(res, 0, n_out)
            bp2d, _ = wl.result_column(res, 1, n_out)
            pp2d, _ = wl.result_column(res, 2, n_out)
            key = key2d.reshape(-1)[:n_out].to(torch.int64)
            bpay = bp2d.reshape(-1)[:n_out]
            ppay = pp2d.reshape(-1)[:n_out]
            assert bool((bpay % wl.PAY_MUL == 0).all()) and bool((ppay % wl.PAY_MUL == 0).all())
            brow = bpay // wl.PAY_MUL
            assert bool((wl.build_key_of_row(brow, total) == key).all()), "output key differs from its build row's key"
            sum_key += int(key.sum())
            sum_mix = (sum_mix + int(wl._mix64(ppay).sum())) & mask
            want_key += int(rel.sk.to(torch.int64).sum())
            want_mix = (want_mix + int(wl._mix64(rel.sp).sum())) & mask
            res.free()
        assert rows == total
        assert sum_key == want_key and sum_mix == want_mix
    finally:
        for row in tables:
            for t in row:
                t.release()
        ctx.destroy()
        torch.cuda.empty_cache()
