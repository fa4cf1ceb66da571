"""world_size-2 gloo worker for tests/test_dist_gloo.py: runs pyrj.dist.ShardedJoin with a CPU
stand-in for the two local stages (numpy partition + oracle join) so that the exchange logic —
count all-to-all, split sizes, tuple all-to-all — is exercised without a GPU."""
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "radix-join_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import _oracle  # noqa: E402
from pyrj import dist as rjdist  # noqa: E402
from pyrj import hashing, pages as pg, plan as pl  # noqa: E402


class CpuOps:
    """Same interface as pyrj.dist.GpuOps; `table` is a pyrj ColumnarTable."""

    def partition(self, table, n_rows, n_ranks, key_col=0, carry_col=1):
        k, kv = pg.unpack_fixed(table.columns[key_col].pages, n_rows, pl.INT32)
        c, _ = pg.unpack_fixed(table.columns[carry_col].pages, n_rows, pl.INT32)
        k, c = k[kv], c[kv]  # NULL keys never match: dropped in stage A
        dest = hashing.owner_rank(k, n_ranks)
        order = np.argsort(dest, kind="stable")
        counts = np.bincount(dest, minlength=n_ranks).tolist()
        return torch.from_numpy(k[order].copy()), torch.from_numpy(c[order].copy()), counts

    def empty(self, n):
        return torch.empty(n, dtype=torch.int32)

    def join(self, bk, bc, pk, pc, skip_rank_bits):
        p = pl.Plan()
        p.new_scan_node(0, [(0, pl.INT32), (1, pl.INT32)])
        p.new_scan_node(1, [(0, pl.INT32), (1, pl.INT32)])
        p.new_join_node(True, 0, 1, 0, 0, [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)])
        p.new_input(pl.make_table([(pl.INT32, bk.numpy()), (pl.INT32, bc.numpy())]))
        p.new_input(pl.make_table([(pl.INT32, pk.numpy()), (pl.INT32, pc.numpy())]))
        p.root = 2
        return _oracle.execute(p), bk.numpy().copy(), pk.numpy().copy()


def main():
    out_dir = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(100 + rank)
    nb, npr = 3000 + 500 * rank, 7000 - 900 * rank  # ragged shards
    total = 8000
    bk = rng.integers(0, total, nb).astype(np.int32)
    bv = rng.random(nb) > 0.05
    pk = rng.integers(0, total, npr).astype(np.int32)
    pv = rng.random(npr) > 0.05
    bt = pl.make_table([(pl.INT32, bk, bv), (pl.INT32, np.arange(nb, dtype=np.int32) + 100000 * rank)])
    pt = pl.make_table([(pl.INT32, pk, pv), (pl.INT32, np.arange(npr, dtype=np.int32) + 100000 * rank)])
    sj = rjdist.ShardedJoin(CpuOps())
    res, got_bk, got_pk = sj.run(bt, nb, pt, npr)
    with open(os.path.join(out_dir, f"rank{rank}.pkl"), "wb") as f:
        pickle.dump({
            "rows": pl.table_rows(res), "build_keys": got_bk, "probe_keys": got_pk,
            "shard": (bk, bv, pk, pv, rank, nb, npr),
        }, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
