"""bench.py, N > 1: one process per GPU, the join sharded by key hash with one exchange step.

Launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`; bench.py
hands over to main() below.  See bench.py's docstring for the contract.
"""
import json
import os
import time

import torch
import torch.distributed as dist

from pyrj import capi
from pyrj import workloads as wl


def main(args, rank, world, dev_index, device):
    # RJ_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    # (ranks share devices, the exchange is staged through the host); the default is RCCL.
    backend = os.environ.get("RJ_BENCH_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)

    from pyrj import dist as rjdist

    name = "uniform1b"
    total = args.rows or wl.WORKLOADS[name]["rows"]
    ctx = capi.Context(device=dev_index, profile=False)
    rel = wl.make_relations(name, device, rows=total, rank=rank, world=world)
    R = wl.adopt(ctx, [rel.rk, rel.rp])
    S = wl.adopt(ctx, [rel.sk, rel.sp])
    n = rel.n
    torch.cuda.empty_cache()
    sj = rjdist.ShardedJoin(rjdist.GpuOps(ctx, device))

    def step():
        res = sj.run(R, n, S, n)
        rows = res.num_rows
        res.free()
        return rows

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    rows = 0
    for _ in range(args.warmup):
        rows = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows = step()
    barrier()
    dt = time.perf_counter() - t0

    rdev = device if backend == "nccl" else torch.device("cpu")
    tt = torch.tensor([dt], dtype=torch.float64, device=rdev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    tr = torch.tensor([rows], dtype=torch.int64, device=rdev)
    dist.all_reduce(tr, op=dist.ReduceOp.SUM)
    total_rows = int(tr.item())
    # every probe key hits exactly one build row: |out| = |S|
    if total_rows != total:
        raise SystemExit(f"wrong result size: {total_rows} != {total}")
    if rank == 0:
        info = ctx.device_info()
        out = {
            "metric": "probe_tuples_per_sec",
            "value": total * args.steps / dt,
            "unit": "tuples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": wl.WORKLOADS[name]["label"] + f"; {total} rows per relation overall, 1/{world} of both relations per GPU",
                "rows_per_relation": total,
                "parallelism": f"hash-sharded x{world}, one all-to-all ({backend})",
                "device": info["name"],
                "arch": info["arch"],
            },
            "roofline": None,
        }
        print(json.dumps(out))
    R.release()
    S.release()
    ctx.destroy()
    dist.destroy_process_group()
