"""Child process of tests/test_gpu_sharded.py: the failure paths of the RCCL transport on one GPU.
  inject N   a world-1 RCCL job in which this rank fails locally at point N of the sharded join
             (RJ_DEBUG_SHARD_FAIL: 1 preparing, 2 stage A, 3 receive buffers): the status word that
             travels with the counts makes it give up BEFORE the exchange, within seconds; a fresh
             context of the same process then joins correctly;
  stall      the same job, but the rank's stage A takes "forever" (RJ_DEBUG_SHARD_FAIL=4: 7 s) with
             RJ_EXCHANGE_TIMEOUT_MS=3000: the wait for the exchange expires, the communicator is
             aborted, the call returns RJ_ERR_DEVICE, later calls fail at once, destroy returns;
  bringup    a context that claims to be rank 0 of a TWO-rank job whose rank 1 never starts:
             communicator bring-up is bounded by RJ_EXCHANGE_TIMEOUT_MS and fails instead of hanging.
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "radix-join_amd"), HERE]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
mode = sys.argv[1]
if mode == "inject":
    os.environ["RJ_DEBUG_SHARD_FAIL"] = sys.argv[2]
    os.environ["RJ_DEBUG_SHARD_FAIL_RANK"] = "0"
elif mode == "stall":
    os.environ["RJ_DEBUG_SHARD_FAIL"] = "4"
    os.environ["RJ_DEBUG_SHARD_FAIL_RANK"] = "0"
    os.environ["RJ_EXCHANGE_TIMEOUT_MS"] = "3000"
    os.environ["RJ_BRINGUP_TIMEOUT_MS"] = "60000"  # (a cold process takes seconds to load RCCL)
else:
    os.environ["RJ_EXCHANGE_TIMEOUT_MS"] = "4000"

import numpy as np  # noqa: E402

import _oracle  # noqa: E402
from pyrj import capi  # noqa: E402
from pyrj import plan as pl  # noqa: E402

cid = capi.make_comm_id()
if mode == "bringup":
    t0 = time.time()
    try:
        capi.Context(devices=[0], world_size=2, rank_base=0, comm_id=cid, exchange=capi.EXCHANGE_RCCL)
    except capi.RjError as e:
        dt = time.time() - t0
        assert e.code == 2, e  # RJ_ERR_DEVICE
        assert "bring-up" in e.message and "RJ_EXCHANGE_TIMEOUT_MS" in e.message, e.message
        assert 3.0 < dt < 30.0, dt
        print(f"bring-up without its peer failed after {dt:.1f} s: {e.message}", flush=True)
        # a stuck helper thread may still sit inside RCCL's bootstrap: leave without running
        # destructors that would wait for it
        sys.stdout.flush()
        os._exit(0)
    raise SystemExit("bring-up of a two-rank job with one rank did not fail")

rng = np.random.default_rng(31)
nb, npr = 200_000, 400_000
p = pl.Plan()
p.new_scan_node(0, [(0, pl.INT32), (1, pl.INT32)])
p.new_scan_node(1, [(0, pl.INT32), (1, pl.INT32)])
p.new_join_node(True, 0, 1, 0, 0, [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)])
p.new_input(pl.make_table([(pl.INT32, rng.permutation(nb).astype(np.int32)), (pl.INT32, np.arange(nb, dtype=np.int32))]))
p.new_input(pl.make_table([(pl.INT32, rng.integers(0, nb, npr).astype(np.int32)), (pl.INT32, np.arange(npr, dtype=np.int32))]))
p.root = 2
ctx = capi.Context(devices=[0], world_size=1, rank_base=0, comm_id=cid, exchange=capi.EXCHANGE_RCCL)
tables = [[ctx.lane(0).upload(t) for t in p.inputs]]
t0 = time.time()
if mode == "stall":
    try:
        ctx.execute_sharded(p, tables)
    except capi.RjError as e:
        dt = time.time() - t0
        assert e.code == 2 and "did not complete within 3000 ms" in e.message, e
        assert 2.5 < dt < 15.0, dt  # (the error surfaces once this rank's own queued kernels have drained)
        print(f"stalled exchange gave up after {dt:.1f} s: {e.message}", flush=True)
    else:
        raise SystemExit(f"the stalled exchange did not time out (the join returned after {time.time() - t0:.2f} s)")
    try:  # the transport is marked failed: nothing else may be tried on it
        ctx.execute_sharded(p, tables)
    except capi.RjError as e:
        assert "failed earlier" in e.message, e
    else:
        raise SystemExit("a failed transport accepted another join")
    t1 = time.time()
    for t in tables[0]:
        t.release()
    ctx.destroy()
    print(f"destroyed {time.time() - t1:.1f} s later", flush=True)
    sys.exit(0)
try:
    ctx.execute_sharded(p, tables)
except capi.RjError as e:
    dt = time.time() - t0
    assert e.code == 3 and "injected failure" in e.message, e  # the failing rank rethrows its own error
    assert dt < 10.0, dt
    print(f"injected failure {sys.argv[2]} surfaced after {dt * 1e3:.0f} ms: {e.message}", flush=True)
else:
    raise SystemExit("the injected failure did not surface")
# nothing of the failed join lingers: the same context joins correctly afterwards
os.environ.pop("RJ_DEBUG_SHARD_FAIL")
ctx.destroy()
ctx = capi.Context(devices=[0], world_size=1, rank_base=0, comm_id=capi.make_comm_id(), exchange=capi.EXCHANGE_RCCL)
tables = [[ctx.lane(0).upload(t) for t in p.inputs]]
(res,) = ctx.execute_sharded(p, tables)
got = res.to_table()
res.free()
want = _oracle.execute(p)
assert got.num_rows == want.num_rows and pl.table_digest(got) == pl.table_digest(want)
for t in tables[0]:
    t.release()
ctx.destroy()
print("a fresh communicator joins correctly afterwards", flush=True)
