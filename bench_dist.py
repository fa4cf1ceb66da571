"""bench.py, N > 1: one process per GPU, the join sharded by key hash INSIDE the library
(rj_execute_sharded: stage A by owner rank, one RCCL all-to-all per relation on an exchange
stream, stage B = local radix passes + build/probe).

Launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`; bench.py
hands over to main() below.  STRONG scaling: the job is BASELINE config 3 (1B ⋈ 1B INT32 keys,
Zipf-0.9 probe keys, INT64 payloads — the same job bench.py runs at N = 1, and BASELINE config 4
at N = 8); every rank holds 1/N of the rows of both relations.  torch.distributed only carries the
128-byte communicator id to the ranks, the barrier around the timed region and the final
reductions; the data path collective is the library's own.
"""
import json
import os
import time

import torch
import torch.distributed as dist

from pyrj import capi
from pyrj import workloads as wl


def verify_sharded(res, rel, device):
    """This rank's slice of the result, in closed form: the build row an output row names holds
    its key (the build side's keys are a known bijection of the global row index); globally, the
    slices add up to |S| rows, to the probe side's key sum and payload checksum."""
    n_out = res.num_rows
    mul = wl.PAY_MUL if rel.payload64 else 1
    rf_pay = wl.ROWS64 if rel.payload64 else wl.ROWS32
    sum_key = 0
    sum_mix = 0
    if n_out:
        key2d, _ = wl.result_column(res, 0, n_out)
        bp2d, _ = wl.result_column(res, 1, n_out)
        pp2d, _ = wl.result_column(res, 2, n_out)
        step = 1 << 26
        for r0 in range(0, n_out, step):
            r1 = min(n_out, r0 + step)

            def rows_of(col2d, rf):
                p0, p1 = r0 // rf, (r1 + rf - 1) // rf
                return col2d[p0:p1].reshape(-1)[r0 - p0 * rf : r1 - p0 * rf]

            key = rows_of(key2d, wl.ROWS32).to(torch.int64)
            bpay = rows_of(bp2d, rf_pay).to(torch.int64)
            ppay = rows_of(pp2d, rf_pay).to(torch.int64)
            assert bool((bpay % mul == 0).all()) and bool((ppay % mul == 0).all()), "payload is not a row multiple"
            brow = bpay // mul
            assert bool(((brow >= 0) & (brow < rel.total)).all()), "build row id out of range"
            assert bool((wl.build_key_of_row(brow, rel.total) == key).all()), "output key differs from its build row's key"
            sum_key += int(key.sum())
            sum_mix = (sum_mix + int(wl._mix64(ppay).sum())) & 0xFFFFFFFFFFFFFFFF
    want_key = int(rel.sk.to(torch.int64).sum())
    want_mix = 0
    for i in range(0, rel.n, 1 << 27):
        want_mix = (want_mix + int(wl._mix64(rel.sp[i : i + (1 << 27)].to(torch.int64)).sum())) & 0xFFFFFFFFFFFFFFFF
    # int64 all-reduce wraps like the checksums do
    def s64(x):
        x &= 0xFFFFFFFFFFFFFFFF
        return x - (1 << 64) if x >= (1 << 63) else x

    t = torch.tensor([n_out, s64(sum_key), s64(want_key), s64(sum_mix), s64(want_mix)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    rows, gk, wk, gm, wm = [int(v) for v in t.tolist()]
    assert rows == rel.total, f"result has {rows} rows over all ranks, expected |S| = {rel.total}"
    assert gk == wk, "sum of output keys differs from the sum of probe keys"
    assert gm == wm, "checksum of output probe payloads differs from the probe side's"
    return {"rows": rows, "sum_key": gk & 0xFFFFFFFFFFFFFFFF, "sum_mix_probe_payload": gm & 0xFFFFFFFFFFFFFFFF}


def main(args, rank, world, dev_index, device):
    # RCCL prints a version banner on stdout when a communicator comes up; stdout is reserved for
    # the ONE JSON line, so everything else of this run goes to stderr at the descriptor level
    import sys

    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=device)
    name = args.workload
    total = args.rows or wl.WORKLOADS[name]["rows"]

    # rank 0 makes the communicator id; everyone gets its 128 bytes
    cid = torch.zeros(128, dtype=torch.uint8, device=device)
    if rank == 0:
        cid = torch.frombuffer(bytearray(capi.make_comm_id()), dtype=torch.uint8).to(device)
    dist.broadcast(cid, src=0)
    ctx = capi.Context(devices=[dev_index], world_size=world, rank_base=rank, comm_id=bytes(cid.cpu().numpy().tobytes()),
                       exchange=capi.EXCHANGE_RCCL, profile=True)
    lane = ctx.lane(0)
    rel = wl.make_relations(name, device, rows=total, rank=rank, world=world, sharded=True)
    R = wl.adopt(lane, [rel.rk, rel.rp])
    S = wl.adopt(lane, [rel.sk, rel.sp])
    torch.cuda.empty_cache()
    plan = wl.join_plan(rel.payload_type)

    def step(keep=False):
        (res,) = ctx.execute_sharded(plan, [[R, S]])
        if keep:
            return res
        res.free()

    def barrier():
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    last = None
    for i in range(args.steps):
        if i + 1 == args.steps:
            last = step(keep=True)
        else:
            step()
    barrier()
    dt = time.perf_counter() - t0
    stats = ctx.profile()

    tt = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    digest = None
    if not args.no_verify:
        try:
            digest = verify_sharded(last, rel, device)
        except AssertionError as e:
            raise SystemExit(f"rank {rank}: RESULT VERIFICATION FAILED: {e}")
    else:
        tr = torch.tensor([last.num_rows], dtype=torch.int64, device=device)
        dist.all_reduce(tr, op=dist.ReduceOp.SUM)
        if int(tr.item()) != total:
            raise SystemExit(f"wrong result size: {int(tr.item())} != {total}")
    last.free()
    if rank == 0:
        import bench

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
                "workload": wl.WORKLOADS[name]["label"] + f"; {total} rows per relation overall, 1/{world} of both relations per GPU, Page-packed in HBM",
                "rows_per_relation": total,
                "parallelism": f"hash-sharded x{world}: stage A by owner rank, one RCCL all-to-all per relation inside librj, local radix passes + build/probe",
                "device": info["name"],
                "arch": info["arch"],
                "verified": digest,
            },
            # redistribution as rank 0 saw it: ms per step on its exchange stream, from "my slices
            # are ready" to "everything has left / arrived" (includes waiting for slower peers; it
            # overlaps the probe side's stage-A scatter and the build side's local passes)
            "exchange_ms": {s["name"]: s["total_ms"] / args.steps for s in (stats or []) if s["name"].startswith("exchange")},
            # rank 0's kernels on its 1/N share (the exchange itself is not a kernel of ours)
            "roofline": bench.roofline(stats, rel.n, rel.n, args.steps, 8 if rel.payload64 else 4, f"{name}_x{world}") if stats else None,
        }
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    R.release()
    S.release()
    ctx.destroy()
    dist.destroy_process_group()
