"""The multi-GPU join behind the C-ABI (rj_execute_sharded / a context that owns several
devices), exercised on ONE GPU: a context over devices [0, 0, ...] runs N virtual ranks with the
real stage A / exchange / stage B code (the peer-copy transport degenerates to device-to-device
copies), and the RCCL transport is run at world size 1 (self send/recv).  Results of all ranks
together must equal the oracle's join of the unsharded inputs (multiset digest).  Real multi-GPU
runs are the driver's (bench.py --gpus N)."""
import numpy as np
import pytest

import _oracle
from pyrj import capi
from pyrj import pages as pg
from pyrj import plan as pl

pytestmark = pytest.mark.gpu


def shard_table(t: pl.ColumnarTable, n_ranks: int):
    """Contiguous row shards of a fixed-width table (decode + re-encode on the host)."""
    cols = [pg.unpack_fixed(c.pages, t.num_rows, c.type) for c in t.columns]
    cuts = [t.num_rows * r // n_ranks for r in range(n_ranks + 1)]
    out = []
    for r in range(n_ranks):
        a, b = cuts[r], cuts[r + 1]
        out.append(pl.make_table([(c.type, v[a:b], m[a:b]) for c, (v, m) in zip(t.columns, cols)]))
    return out


def combine(digests):
    n = sum(d[0] for d in digests)
    s = sum(d[1] for d in digests) & 0xFFFFFFFFFFFFFFFF
    x = 0
    for d in digests:
        x ^= d[2]
    return (n, s, x)


def run_sharded(plan, n_ranks, xcd_placement=False, **ctx_kw):
    import os

    # xcd_placement: force the XCD-aware output placement of the big passes (normally from 40 Mi
    # tuples) onto these small shards — stage A and the passes behind the exchange then run it
    old = os.environ.get("RJ_TUNE_XCD_MIN_ROWS")
    if xcd_placement:
        os.environ["RJ_TUNE_XCD_MIN_ROWS"] = "0"  # read once, when the context is created
    try:
        ctx = capi.Context(devices=[0] * n_ranks, **ctx_kw)
    finally:
        if xcd_placement:
            if old is None:
                del os.environ["RJ_TUNE_XCD_MIN_ROWS"]
            else:
                os.environ["RJ_TUNE_XCD_MIN_ROWS"] = old
    tables = []
    try:
        assert ctx.n_devices == n_ranks
        shards = [shard_table(t, n_ranks) for t in plan.inputs]
        tables = [[ctx.lane(d).upload(shards[i][d]) for i in range(len(plan.inputs))] for d in range(n_ranks)]
        res = ctx.execute_sharded(plan, tables)
        out = [r.to_table() for r in res]
        for r in res:
            r.free()
        return out
    finally:
        for row in tables:
            for t in row:
                t.release()
        ctx.destroy()


def check_against_oracle(plan, n_ranks, **ctx_kw):
    parts = run_sharded(plan, n_ranks, **ctx_kw)
    want = _oracle.execute(plan)
    assert sum(p.num_rows for p in parts) == want.num_rows
    for p in parts:
        assert [c.type for c in p.columns] == [c.type for c in want.columns]
    assert combine([pl.table_digest(p) for p in parts if p.num_rows]) == pl.table_digest(want)
    return parts


def join_plan(bt, pt, btypes, ptypes, outs, build_left=True):
    p = pl.Plan()
    if build_left:
        p.new_scan_node(0, list(enumerate(btypes)))
        p.new_scan_node(1, list(enumerate(ptypes)))
        p.new_join_node(True, 0, 1, 0, 0, outs)
        p.new_input(bt)
        p.new_input(pt)
    else:
        p.new_scan_node(0, list(enumerate(ptypes)))
        p.new_scan_node(1, list(enumerate(btypes)))
        p.new_join_node(False, 0, 1, 0, 0, outs)
        p.new_input(pt)
        p.new_input(bt)
    p.root = 2
    return p


@pytest.mark.parametrize("xcd", [False, True])
@pytest.mark.parametrize("n_ranks", [2, 4, 8])
def test_baseline_shape_int32_payloads(n_ranks, xcd):
    """config-2/4 shape: unique build keys, uniform probe keys, INT32 payloads (packed pairs:
    ONE array moves per relation)"""
    rng = np.random.default_rng(10 + n_ranks)
    nb, npr = 1_500_000, 2_500_000
    bt = pl.make_table([(pl.INT32, rng.permutation(nb).astype(np.int32)), (pl.INT32, np.arange(nb, dtype=np.int32))])
    pt = pl.make_table([(pl.INT32, rng.integers(0, nb + 100_000, npr).astype(np.int32)), (pl.INT32, np.arange(npr, dtype=np.int32))])
    parts = check_against_oracle(join_plan(bt, pt, [pl.INT32, pl.INT32], [pl.INT32, pl.INT32], [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)]), n_ranks, xcd_placement=xcd)
    # every rank owns a share of the result (hash sharding balances uniform keys)
    assert all(p.num_rows > 0.5 * sum(q.num_rows for q in parts) / n_ranks for p in parts)


@pytest.mark.parametrize("xcd", [False, True])
@pytest.mark.parametrize("build_left", [True, False])
def test_int64_payloads_null_keys_duplicates(build_left, xcd):
    """config-3 shape at small scale: INT64 payloads on both sides (12-byte tuples: a key array
    and a pair array move per relation), NULL keys on both sides, duplicate build keys, a hot
    probe key"""
    rng = np.random.default_rng(21)
    nb, npr = 400_000, 900_000
    bk = rng.integers(0, 300_000, nb).astype(np.int32)
    pk = rng.integers(0, 330_000, npr).astype(np.int32)
    pk[rng.random(npr) < 0.05] = 4242
    bt = pl.make_table([(pl.INT32, bk, rng.random(nb) > 0.03), (pl.INT64, rng.integers(-(2**62), 2**62, nb).astype(np.int64))])
    pt = pl.make_table([(pl.INT32, pk, rng.random(npr) > 0.02), (pl.INT64, rng.integers(-(2**62), 2**62, npr).astype(np.int64))])
    if build_left:
        outs = [(0, pl.INT32), (1, pl.INT64), (3, pl.INT64)]
    else:
        outs = [(3, pl.INT64), (0, pl.INT32), (1, pl.INT64)]
    check_against_oracle(join_plan(bt, pt, [pl.INT32, pl.INT64], [pl.INT32, pl.INT64], outs, build_left), 4, xcd_placement=xcd)


def test_int64_keys_and_key_only_outputs():
    """KW = 2 (INT64 keys), one side without payload, the other with an INT32 payload"""
    rng = np.random.default_rng(22)
    nb, npr = 200_000, 500_000
    bk = (rng.integers(0, 150_000, nb).astype(np.int64) * 4_000_000_007) - 99
    pk = rng.choice(bk, npr)
    bt = pl.make_table([(pl.INT64, bk)])
    pt = pl.make_table([(pl.INT64, pk), (pl.INT32, np.arange(npr, dtype=np.int32))])
    p = pl.Plan()
    p.new_scan_node(0, [(0, pl.INT64)])
    p.new_scan_node(1, [(0, pl.INT64), (1, pl.INT32)])
    p.new_join_node(True, 0, 1, 0, 0, [(1, pl.INT64), (2, pl.INT32)])
    p.new_input(bt)
    p.new_input(pt)
    p.root = 2
    check_against_oracle(p, 2)


def test_forced_radix_bits_reach_into_the_rank_bits():
    """rj_config.radix_bits = 21 with 8 ranks: 21 radix bits + 3 rank bits leave 8 varying bits
    above the radix digits.  The rank bits are constant on a rank; bucket indices come from the
    LOW bits above the radix digits — exactly the ones that still vary — so every distinct key of a
    partition keeps its own home bucket.  Results must not depend on any of it."""
    rng = np.random.default_rng(23)
    nb, npr = 300_000, 600_000
    bt = pl.make_table([(pl.INT32, rng.integers(0, 250_000, nb).astype(np.int32)), (pl.INT32, np.arange(nb, dtype=np.int32))])
    pt = pl.make_table([(pl.INT32, rng.integers(0, 260_000, npr).astype(np.int32)), (pl.INT32, np.arange(npr, dtype=np.int32))])
    check_against_oracle(join_plan(bt, pt, [pl.INT32, pl.INT32], [pl.INT32, pl.INT32], [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)]), 8, radix_bits=21)


def test_two_joins_stay_sharded():
    """(R ⋈ S) ⋈ T: the intermediate result stays distributed and is re-sharded by the parent's key"""
    rng = np.random.default_rng(24)
    r = pl.make_table([(pl.INT32, rng.permutation(100_000).astype(np.int32)), (pl.INT32, rng.integers(0, 50_000, 100_000).astype(np.int32))])
    s = pl.make_table([(pl.INT32, rng.integers(0, 100_000, 300_000).astype(np.int32))])
    t = pl.make_table([(pl.INT32, rng.integers(0, 50_000, 80_000).astype(np.int32)), (pl.INT64, rng.integers(0, 2**40, 80_000).astype(np.int64))])
    p = pl.Plan()
    a = p.new_scan_node(0, [(0, pl.INT32), (1, pl.INT32)])
    b = p.new_scan_node(1, [(0, pl.INT32)])
    j1 = p.new_join_node(True, a, b, 0, 0, [(0, pl.INT32), (1, pl.INT32)])  # -> (r.key, r.fk)
    c = p.new_scan_node(2, [(0, pl.INT32), (1, pl.INT64)])
    j2 = p.new_join_node(False, j1, c, 1, 0, [(0, pl.INT32), (3, pl.INT64), (1, pl.INT32)])
    for x in (r, s, t):
        p.new_input(x)
    p.root = j2
    check_against_oracle(p, 4)


def test_empty_sides_and_empty_shards():
    rng = np.random.default_rng(25)
    bt = pl.make_table([(pl.INT32, np.arange(5, dtype=np.int32)), (pl.INT32, np.arange(5, dtype=np.int32))])  # fewer rows than ranks
    pt = pl.make_table([(pl.INT32, rng.integers(0, 7, 1000).astype(np.int32)), (pl.INT32, np.arange(1000, dtype=np.int32))])
    check_against_oracle(join_plan(bt, pt, [pl.INT32, pl.INT32], [pl.INT32, pl.INT32], [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)]), 8)
    empty = pl.make_table([(pl.INT32, np.zeros(0, np.int32)), (pl.INT32, np.zeros(0, np.int32))])
    parts = check_against_oracle(join_plan(empty, pt, [pl.INT32, pl.INT32], [pl.INT32, pl.INT32], [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)]), 2)
    assert all(p.num_rows == 0 and all(c.pages.shape[0] == 0 for c in p.columns) for p in parts)


def test_unshardable_plans_are_refused_loudly():
    rng = np.random.default_rng(26)
    n = 10_000
    pt = pl.make_table([(pl.INT32, rng.integers(0, n, n).astype(np.int32)), (pl.INT32, np.arange(n, dtype=np.int32))])
    p = pl.Plan()
    bt = pl.make_table([(pl.INT32, rng.permutation(n).astype(np.int32)), (pl.INT64, np.arange(n, dtype=np.int64)), (pl.INT64, np.arange(n, dtype=np.int64))])
    p.new_scan_node(0, [(0, pl.INT32), (1, pl.INT64), (2, pl.INT64)])
    p.new_scan_node(1, [(0, pl.INT32), (1, pl.INT32)])
    p.new_join_node(True, 0, 1, 0, 0, [(0, pl.INT32), (1, pl.INT64), (2, pl.INT64), (4, pl.INT32)])  # two INT64 payloads of one side: 4 carry words
    p.new_input(bt)
    p.new_input(pt)
    p.root = 2
    assert not capi.plan_shardable(p)[0]
    with pytest.raises(capi.RjError) as e:
        run_sharded(p, 2)
    assert e.value.code == 5 and "row index" in e.value.message


@pytest.mark.parametrize("n_ranks", [2, 8])
def test_wide_carries_shard(n_ranks):
    """Several payload columns per side and NULL-bearing ones travel with the key (up to three carry
    words, validity bits in a word of their own), so such joins shard: INT32 + INT64 on the build
    side, two INT32 (one with NULLs) on the probe side, NULL keys, duplicates — and a second join
    on top whose build side is that (distributed) result."""
    rng = np.random.default_rng(60 + n_ranks)
    nb, npr = 500_000, 1_100_000
    bt = pl.make_table([
        (pl.INT32, rng.integers(0, 400_000, nb).astype(np.int32), rng.random(nb) > 0.02),
        (pl.INT32, rng.integers(-(2**31), 2**31 - 1, nb).astype(np.int32)),
        (pl.INT64, rng.integers(-(2**62), 2**62, nb).astype(np.int64)),
    ])
    pt = pl.make_table([
        (pl.INT32, rng.integers(0, 420_000, npr).astype(np.int32)),
        (pl.INT32, rng.integers(0, 1000, npr).astype(np.int32), rng.random(npr) > 0.3),
        (pl.INT32, np.arange(npr, dtype=np.int32)),
    ])
    tt = pl.make_table([(pl.INT32, rng.integers(0, 1000, 3000).astype(np.int32)), (pl.FP64, rng.standard_normal(3000))])
    p = pl.Plan()
    a = p.new_scan_node(0, [(0, pl.INT32), (1, pl.INT32), (2, pl.INT64)])
    b = p.new_scan_node(1, [(0, pl.INT32), (1, pl.INT32), (2, pl.INT32)])
    j1 = p.new_join_node(True, a, b, 0, 0, [(2, pl.INT64), (4, pl.INT32), (1, pl.INT32), (5, pl.INT32), (0, pl.INT32)])
    c = p.new_scan_node(2, [(0, pl.INT32), (1, pl.FP64)])
    # probe side of the second join = the first join's result, keyed by its NULL-bearing column
    j2 = p.new_join_node(False, j1, c, 1, 0, [(0, pl.INT64), (6, pl.FP64), (3, pl.INT32), (1, pl.INT32)])
    for x in (bt, pt, tt):
        p.new_input(x)
    p.root = j2
    assert capi.plan_shardable(p)[0]
    check_against_oracle(p, n_ranks)


@pytest.mark.parametrize("n_ranks", [2, 4])
def test_nulls_on_one_rank_only(n_ranks):
    """Whether a payload column needs a validity word depends on the data of a shard: here the build
    side's only payload column holds NULLs in the FIRST rank's rows alone (elsewhere it could travel
    as it is, CARRY_COLUMN), and one of the probe side's two columns in the LAST rank's rows alone
    (elsewhere two words instead of three).  The ranks agree on the union before they cut their
    tuples (Side::null_mask), or the exchanged arrays would not even have the same width."""
    rng = np.random.default_rng(90 + n_ranks)
    nb, npr = 400_000, 900_000
    bvalid = np.ones(nb, dtype=bool)
    bvalid[: nb // (2 * n_ranks)] = rng.random(nb // (2 * n_ranks)) > 0.5
    pvalid = np.ones(npr, dtype=bool)
    pvalid[-(npr // (2 * n_ranks)):] = rng.random(npr // (2 * n_ranks)) > 0.4
    bt = pl.make_table([(pl.INT32, rng.integers(0, 300_000, nb).astype(np.int32)),
                        (pl.INT64, rng.integers(-(2**62), 2**62, nb).astype(np.int64), bvalid)])
    pt = pl.make_table([(pl.INT32, rng.integers(0, 320_000, npr).astype(np.int32)),
                        (pl.INT32, rng.integers(-(2**31), 2**31 - 1, npr).astype(np.int32), pvalid),
                        (pl.INT32, np.arange(npr, dtype=np.int32))])
    p = join_plan(bt, pt, [pl.INT32, pl.INT64], [pl.INT32, pl.INT32, pl.INT32],
                  [(1, pl.INT64), (0, pl.INT32), (3, pl.INT32), (4, pl.INT32)])
    assert capi.plan_shardable(p)[0]
    check_against_oracle(p, n_ranks)
    # ... and a single nullable INT32 column on the probe side (one word here, value + validity there)
    p2 = join_plan(bt, pt, [pl.INT32, pl.INT64], [pl.INT32, pl.INT32, pl.INT32], [(0, pl.INT32), (3, pl.INT32)])
    check_against_oracle(p2, n_ranks)


def test_rccl_transport_at_world_size_one():
    """The RCCL code path (communicator from an rj_comm_id, count all-gather, grouped
    ncclSend/ncclRecv to self) on the one GPU this box has.  In a process of its own
    (tests/_rccl_world1.py): RCCL stays out of the test runner, as it stays out of every
    single-GPU deployment of the library."""
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "_rccl_world1.py")], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr[-2000:])
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    assert "rccl world-1 join matches the oracle" in r.stdout


def _child(args, timeout=300):
    import os
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "_rccl_fail.py")] + args, capture_output=True, text=True, timeout=timeout)
    print(r.stdout, r.stderr[-2000:])
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    return r.stdout


@pytest.mark.parametrize("at", [1, 2, 3, 5, 6])
def test_rccl_local_failure_is_agreed_on_before_the_exchange(at):
    """A rank that fails locally (while preparing, in stage A, allocating its receive buffers, or — 5 —
    already in a scan below the join) reports it (6, behind the exchange: it is held back for the next join's
    status word or the end of the plan) in the status word of the next count all-gather; every rank gives up before the
    all-to-all — an error within seconds instead of peers blocked in a collective."""
    out = _child(["inject", str(at)])
    assert "surfaced after" in out and "joins correctly afterwards" in out


def test_rccl_exchange_wait_is_bounded():
    """A rank whose probe-side slices do not become ready (a 7 s stall) against a 3 s bound: the wait
    for the exchange expires, the communicator is aborted, the call returns RJ_ERR_DEVICE, the
    context refuses further joins and can still be destroyed."""
    out = _child(["stall"], timeout=120)
    assert "stalled exchange gave up after" in out and "destroyed" in out


def test_rccl_bring_up_without_its_peers_is_bounded():
    """Rank 0 of a two-rank job whose rank 1 never starts: ncclCommInitRank would wait forever;
    the bring-up runs on a helper thread against RJ_EXCHANGE_TIMEOUT_MS and fails loudly."""
    out = _child(["bringup"], timeout=120)
    assert "bring-up without its peer failed after" in out


@pytest.mark.parametrize("at", [1, 2, 3, 5, 6])
def test_virtual_ranks_local_failure_of_one_rank(at):
    """The same agreement among four virtual ranks of one process: rank 2 fails, the call returns
    its error, nothing hangs, and the context joins correctly afterwards."""
    import os

    rng = np.random.default_rng(40 + at)
    nb, npr = 300_000, 500_000
    bt = pl.make_table([(pl.INT32, rng.permutation(nb).astype(np.int32)), (pl.INT64, rng.integers(0, 2**40, nb).astype(np.int64))])
    pt = pl.make_table([(pl.INT32, rng.integers(0, nb, npr).astype(np.int32)), (pl.INT32, np.arange(npr, dtype=np.int32))])
    plan = join_plan(bt, pt, [pl.INT32, pl.INT64], [pl.INT32, pl.INT32], [(0, pl.INT32), (1, pl.INT64), (3, pl.INT32)])
    os.environ["RJ_DEBUG_SHARD_FAIL"] = str(at)
    os.environ["RJ_DEBUG_SHARD_FAIL_RANK"] = "2"
    try:
        with pytest.raises(capi.RjError) as e:
            run_sharded(plan, 4)
        assert e.value.code == 3 and "injected failure" in e.value.message and "rank 2" in e.value.message
    finally:
        del os.environ["RJ_DEBUG_SHARD_FAIL"], os.environ["RJ_DEBUG_SHARD_FAIL_RANK"]
    check_against_oracle(plan, 4)


def test_rccl_refuses_virtual_ranks_and_owner_digit_can_stay_unfolded():
    """RCCL cannot run two ranks on one device: refused up front (RJ_ERR_ARG), not tried.  And the
    join is the same with stage A partitioning by owner rank only (RJ_TUNE_FOLD_OWNER=0)."""
    import os

    with pytest.raises(capi.RjError) as e:
        capi.Context(devices=[0, 0], exchange=capi.EXCHANGE_RCCL, comm_id=b"\0" * 128)
    assert e.value.code == 1 and "appears twice" in e.value.message
    rng = np.random.default_rng(41)
    nb, npr = 700_000, 900_000
    bt = pl.make_table([(pl.INT32, rng.permutation(nb).astype(np.int32)), (pl.INT32, np.arange(nb, dtype=np.int32))])
    pt = pl.make_table([(pl.INT32, rng.integers(0, nb, npr).astype(np.int32)), (pl.INT64, rng.integers(0, 2**40, npr).astype(np.int64))])
    plan = join_plan(bt, pt, [pl.INT32, pl.INT32], [pl.INT32, pl.INT64], [(0, pl.INT32), (1, pl.INT32), (3, pl.INT64)])
    os.environ["RJ_TUNE_FOLD_OWNER"] = "0"
    try:
        check_against_oracle(plan, 4)
    finally:
        del os.environ["RJ_TUNE_FOLD_OWNER"]
    check_against_oracle(plan, 4, radix_bits=13)  # folded: 4 owners x 2^7 digits in stage A, 2^6 behind the exchange
    # a per-device handle of lanes 1.. is owned by its group: destroying it is a documented no-op
    ctx = capi.Context(devices=[0, 0])
    try:
        ctx.L.rj_context_destroy(ctx.lane(1).h)
        assert b"per-device handle" in ctx.L.rj_last_error(ctx.lane(1).h)
        tables = [[ctx.lane(d).upload(t) for t in shard] for d, shard in enumerate(zip(*[shard_table(t, 2) for t in plan.inputs]))]
        res = ctx.execute_sharded(plan, tables)
        assert sum(r.num_rows for r in res) == _oracle.execute(plan).num_rows
        for r in res:
            r.free()
        for row in tables:
            for t in row:
                t.release()
    finally:
        ctx.destroy()


def test_plain_context_runs_execute_sharded_as_one_rank():
    rng = np.random.default_rng(28)
    n = 200_000
    bt = pl.make_table([(pl.INT32, rng.permutation(n).astype(np.int32)), (pl.INT32, np.arange(n, dtype=np.int32))])
    pt = pl.make_table([(pl.INT32, rng.integers(0, n, 2 * n).astype(np.int32)), (pl.INT32, np.arange(2 * n, dtype=np.int32))])
    plan = join_plan(bt, pt, [pl.INT32, pl.INT32], [pl.INT32, pl.INT32], [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)])
    ctx = capi.Context()
    try:
        tables = [[ctx.lane(0).upload(t) for t in plan.inputs]]
        (res,) = ctx.execute_sharded(plan, tables)
        got = res.to_table()
        res.free()
        assert pl.table_digest(got) == pl.table_digest(_oracle.execute(plan))
    finally:
        ctx.destroy()


def test_contest_execute_semantics_over_several_devices():
    """rj_execute (host pages in, host pages out — what Contest::execute calls) on a context
    that owns several devices: inputs are cut at multiples of 1984 * 1007 rows (a page boundary
    of INT32 and INT64 columns alike), the plan runs sharded, the result is the concatenation of
    the ranks' pages.  A plan that cannot be sharded runs on the first device instead."""
    rng = np.random.default_rng(29)
    nb, npr = 4_500_000, 8_200_000  # (>= 4 cut units of 1984 * 1007 rows: four ranks all get a shard)
    bt = pl.make_table([(pl.INT32, rng.permutation(nb).astype(np.int32)), (pl.INT64, rng.integers(-(2**62), 2**62, nb).astype(np.int64))])
    pt = pl.make_table([(pl.INT32, rng.integers(0, nb + 50_000, npr).astype(np.int32)), (pl.INT32, np.arange(npr, dtype=np.int32))])
    plan = join_plan(bt, pt, [pl.INT32, pl.INT64], [pl.INT32, pl.INT32], [(0, pl.INT32), (1, pl.INT64), (3, pl.INT32)])
    want = pl.table_digest(_oracle.execute(plan))
    for n_ranks in (2, 3 + 1):
        ctx = capi.Context(devices=[0] * n_ranks)
        try:
            got = capi.execute(plan, ctx)
            assert pl.table_digest(got) == want
            # rank slices end in partially filled pages: more pages than a dense encoding needs
            assert got.columns[0].pages.shape[0] >= (got.num_rows + 1983) // 1984
        finally:
            ctx.destroy()
    # VARCHAR payload: not shardable -> the same context answers from its first device
    small = pl.make_table([(pl.INT32, np.arange(1000, dtype=np.int32)), (pl.VARCHAR, [f"s{i}".encode() for i in range(1000)])])
    probe = pl.make_table([(pl.INT32, rng.integers(0, 1200, 5000).astype(np.int32))])
    p = pl.Plan()
    p.new_scan_node(0, [(0, pl.INT32), (1, pl.VARCHAR)])
    p.new_scan_node(1, [(0, pl.INT32)])
    p.new_join_node(True, 0, 1, 0, 0, [(1, pl.VARCHAR), (2, pl.INT32)])
    p.new_input(small)
    p.new_input(probe)
    p.root = 2
    ctx = capi.Context(devices=[0, 0])
    try:
        got = capi.execute(p, ctx)
        assert pl.sorted_rows(got) == pl.sorted_rows(_oracle.execute(p))
    finally:
        ctx.destroy()
