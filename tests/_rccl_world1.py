"""Child process of tests/test_gpu_sharded.py::test_rccl_transport_at_world_size_one: a sharded
join over the RCCL transport at world size 1, against the oracle."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, os.path.join(ROOT, "radix-join_amd"), HERE]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402

import _oracle  # noqa: E402
from pyrj import capi  # noqa: E402
from pyrj import plan as pl  # noqa: E402

rng = np.random.default_rng(27)
nb, npr = 300_000, 700_000
bt = pl.make_table([(pl.INT32, rng.permutation(nb).astype(np.int32)), (pl.INT64, rng.integers(0, 2**50, nb).astype(np.int64))])
pt = pl.make_table([(pl.INT32, rng.integers(0, nb, npr).astype(np.int32)), (pl.INT32, np.arange(npr, dtype=np.int32))])
p = pl.Plan()
p.new_scan_node(0, [(0, pl.INT32), (1, pl.INT64)])
p.new_scan_node(1, [(0, pl.INT32), (1, pl.INT32)])
p.new_join_node(True, 0, 1, 0, 0, [(0, pl.INT32), (1, pl.INT64), (3, pl.INT32)])
p.new_input(bt)
p.new_input(pt)
p.root = 2
cid = capi.make_comm_id()
assert len(cid) == 128
ctx = capi.Context(devices=[0], world_size=1, rank_base=0, comm_id=cid, exchange=capi.EXCHANGE_RCCL)
tables = [[ctx.lane(0).upload(t) for t in p.inputs]]
(res,) = ctx.execute_sharded(p, tables)
got = res.to_table()
res.free()
for t in tables[0]:
    t.release()
ctx.destroy()
want = _oracle.execute(p)
assert got.num_rows == want.num_rows and pl.table_digest(got) == pl.table_digest(want)
print("rccl world-1 join matches the oracle:", got.num_rows, "rows", flush=True)
