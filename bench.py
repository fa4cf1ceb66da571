#!/usr/bin/env python3
"""bench.py — probe-tuples/s of the radix join on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N = 1  workload = BASELINE.json configs[1]: single JoinNode, synthetic 100M ⋈ 100M INT32
       uniform keys, 1 INT32 payload column per side, plan
       Join(build_left=true, Scan(R){0,1}, Scan(S){0,1}, out={0,1,3}).
       One step = one rj_execute_resident() of that plan: Page-packed inputs already in
       HBM, page decode + one histogram and 2 radix scatters per side + build/probe + Page-encoded result in
       HBM, including the result-size read-back.
N > 1  one process per GPU (torch.distributed, backend nccl = RCCL): every rank holds a
       100M ⋈ 100M shard (weak scaling), stage A partitions by rank, ONE all-to-all over
       xGMI re-distributes the tuples, stage B joins locally.  value = all ranks' probe
       tuples / max-over-ranks time.

The JSON line also carries `roofline` (dominant kernel: algorithmic bytes per launch,
SURVEY.md §8d, over its HIP-event duration measured on the launch stream) and, at N = 1,
`cpu_baseline` (the CPU oracle — a port of the reference algorithm — on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this platform needs dmabuf IPC (RCCL fails with
# `hipIpcGetMemHandle: invalid argument` otherwise); the boxes export it, keep it if they do not
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "radix-join_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from pyrj import capi  # noqa: E402
from pyrj import plan as pl  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)
ROWS32 = 1984

# algorithmic bytes per tuple of each kernel (k = 4-byte key, p = 4-byte carry):
# a radix pass costs 3k+2p = 20 B/tuple: k for the histogram read, (k+p) read + (k+p) write for
# the scatter; the probe phase costs (k+pS) + (k+pR) + 12-byte output row = 28 B/probe tuple.
ALGO_BYTES = {
    "pass1_hist": 4.0,
    "pass1_scatter": 16.0,
    "pass2_hist": 4.0,
    "pass2_scatter": 16.0,
    "join_build_probe": 28.0,
}


def pack_pages_gpu(values: torch.Tensor) -> torch.Tensor:
    """INT32 values (device) -> Page images uint8[n_pages, 8192] on the device, no NULLs
    (layout: reference src/build_table.cpp:472-481; 1984 rows per full page)."""
    n = values.numel()
    npages = (n + ROWS32 - 1) // ROWS32
    pages = torch.zeros((npages, 2048), dtype=torch.int32, device=values.device)
    full = n // ROWS32
    if full:
        pages[:full, 1 : 1 + ROWS32] = values[: full * ROWS32].view(full, ROWS32)
        pages[:full, 0] = ROWS32 | (ROWS32 << 16)
        pages[:full, 1986:] = -1  # 248 bitmap bytes, all rows valid
    rem = n - full * ROWS32
    if rem:
        pages[full, 1 : 1 + rem] = values[full * ROWS32 :]
        pages[full, 0] = rem | (rem << 16)
        b = pages[full].view(torch.uint8)
        nb = (rem + 7) // 8
        bm = torch.full((nb,), 255, dtype=torch.uint8, device=values.device)
        if rem % 8:
            bm[-1] = (1 << (rem % 8)) - 1
        b[8192 - nb :] = bm
    return pages.view(torch.uint8).view(npages, 8192)


ROWS64 = 1007


def pack_pages_gpu64(values: torch.Tensor) -> torch.Tensor:
    """INT64 values (device) -> Page images, 1007 rows per full page, values from byte 8
    (layout: reference src/build_table.cpp:515-524)."""
    n = values.numel()
    npages = (n + ROWS64 - 1) // ROWS64
    pages = torch.zeros((npages, 1024), dtype=torch.int64, device=values.device)
    full = n // ROWS64
    if full:
        pages[:full, 1 : 1 + ROWS64] = values[: full * ROWS64].view(full, ROWS64)
    rem = n - full * ROWS64
    if rem:
        pages[full, 1 : 1 + rem] = values[full * ROWS64 :]
    b = pages.view(torch.uint8).view(npages, 8192)
    cnt = torch.full((npages,), ROWS64, dtype=torch.int32, device=values.device)
    if rem:
        cnt[-1] = rem
    hdr = (cnt | (cnt << 16)).view(torch.uint8).view(npages, 4)
    b[:, :4] = hdr
    nbf = (ROWS64 + 7) // 8  # 126 bitmap bytes of a full page, last byte has 7 valid bits
    if full:
        b[:full, 8192 - nbf :] = 255
        b[:full, 8191] = (1 << (ROWS64 % 8)) - 1
    if rem:
        nb = (rem + 7) // 8
        b[full, 8192 - nb :] = 255
        if rem % 8:
            b[full, 8191] = (1 << (rem % 8)) - 1
    return b


def zipf_keys(n_keys, n, s, device, gen):
    """n draws of a Zipf(s) rank over [0, n_keys), scattered through a fixed bijection so that hot
    keys are not numerically adjacent (skew on the probe side only)."""
    w = torch.arange(1, n_keys + 1, device=device, dtype=torch.float64).pow_(-s)
    cdf = torch.cumsum(w, 0)
    del w
    cdf /= cdf[-1].clone()
    out = torch.empty(n, device=device, dtype=torch.int64)
    step = 1 << 27
    for i in range(0, n, step):
        m = min(step, n - i)
        u = torch.rand(m, device=device, dtype=torch.float64, generator=gen)
        out[i : i + m] = torch.searchsorted(cdf, u, right=True).clamp_(max=n_keys - 1)
    del cdf
    return ((out * 7919 + 13) % n_keys).to(torch.int32)


def make_relations(n, rank, world, device):
    """R: unique keys (a permutation of [0, world*n)), payload = global row index.
    S: uniform iid keys over R's domain, payload = global row index."""
    g = torch.Generator(device=device)
    total = n * world
    if world == 1:
        g.manual_seed(1)
        rk = torch.randperm(n, generator=g, device=device, dtype=torch.int64).to(torch.int32)
    else:
        # multiplicative bijection of [0, total): A is coprime to total = 2^a * 5^b * world
        idx = torch.arange(rank * n, (rank + 1) * n, device=device, dtype=torch.int64)
        rk = ((idx * 2654435761 + 12345) % total).to(torch.int32)
    g.manual_seed(2 + rank)
    sk = torch.randint(0, total, (n,), generator=g, device=device, dtype=torch.int64).to(torch.int32)
    pay = torch.arange(rank * n, (rank + 1) * n, device=device, dtype=torch.int64).to(torch.int32)
    return rk, pay, sk, pay.clone()


def adopt(ctx, cols):
    pages = [pack_pages_gpu64(c) if c.dtype == torch.int64 else pack_pages_gpu(c) for c in cols]
    types = [pl.INT64 if c.dtype == torch.int64 else pl.INT32 for c in cols]
    torch.cuda.synchronize()
    n = cols[0].numel()
    return ctx.adopt_device(n, types, [p.data_ptr() for p in pages], [p.shape[0] for p in pages], keep=pages)


def join_plan(payload=None):
    payload = pl.INT32 if payload is None else payload
    p = pl.Plan()
    p.new_scan_node(0, [(0, pl.INT32), (1, payload)])
    p.new_scan_node(1, [(0, pl.INT32), (1, payload)])
    p.new_join_node(True, 0, 1, 0, 0, [(0, pl.INT32), (1, payload), (3, payload)])
    p.root = 2
    return p


def cpu_baseline(sample_rows):
    """The CPU oracle (port of the reference's execute path) on a bounded sample of the same
    workload, on this box's host cores (single thread: the reference's 8-thread run was slower
    than 1 thread)."""
    import _oracle

    rng = np.random.default_rng(1)
    n = sample_rows
    rk = rng.permutation(n).astype(np.int32)
    sk = rng.integers(0, n, n).astype(np.int32)
    pay = np.arange(n, dtype=np.int32)
    p = join_plan()
    p.new_input(pl.make_table([(pl.INT32, rk), (pl.INT32, pay)]))
    p.new_input(pl.make_table([(pl.INT32, sk), (pl.INT32, pay)]))
    _oracle.lib()
    t0 = time.perf_counter()
    res = _oracle.execute(p)
    dt = time.perf_counter() - t0
    assert res.num_rows == n
    return {
        "value": n / dt,
        "unit": "probe tuples/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{n} x {n} INT32 uniform, same plan, oracle/rjo_oracle.c incl. page decode+encode, {dt:.2f} s",
    }


def roofline(stats, rows_build, rows_probe, steps_profiled, payload_bytes=4):
    # SURVEY.md §8d with p = payload bytes: scatter 2(k+p), probe (k+p)+(k+p)+(k+2p)
    algo_bytes = dict(ALGO_BYTES)
    algo_bytes["pass1_scatter"] = algo_bytes["pass2_scatter"] = 2.0 * (4 + payload_bytes)
    algo_bytes["join_build_probe"] = 3.0 * 4 + 4.0 * payload_bytes
    per = {}
    for s in stats:
        if s["name"] in algo_bytes and s["launches"]:
            per[s["name"]] = s
    if not per:
        return None
    dom = max(per.values(), key=lambda s: s["total_ms"])
    name = dom["name"]
    avg_ms = dom["total_ms"] / dom["launches"]
    if name == "join_build_probe":
        tuples = rows_probe
    else:
        # partition kernels run once per relation per step with equal cardinalities here
        tuples = (rows_build + rows_probe) / 2.0
    algo = algo_bytes[name] * tuples
    achieved = algo / (avg_ms * 1e-3) / 1e9
    # HBM bytes per launch from the PMC passes committed under profiles/ (bench.py cannot
    # collect hardware counters itself); only valid for the size they were collected at
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_r01.json")) as f:
            t = json.load(f)
        if t["rows_per_relation"] == rows_probe == rows_build and payload_bytes == 4:
            traffic = t["bytes_per_launch"].get(name)
    except Exception:
        traffic = None
    return {
        "kernel": name,
        "bound": "hbm",
        "achieved": achieved,
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBPS,
        "traffic": traffic,
        "traffic_source": "profiles/traffic_r01.json (rocprofv3 PMC, gfx950-corrected)" if traffic else None,
        "avg_launch_ms": avg_ms,
        "algorithmic_bytes_per_launch": algo,
        "kernels_ms_per_step": {s["name"]: s["total_ms"] / steps_profiled for s in stats},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=100_000_000, help="rows per relation per GPU")
    ap.add_argument("--cpu-sample", type=int, default=2_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--zipf", type=float, default=0.0, help="Zipf exponent of the probe keys (config 3: 0.9); N=1 only")
    ap.add_argument("--payload64", action="store_true", help="INT64 payload columns (config 3); N=1 only")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    if args.gpus != world and distributed:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the product, there is no CPU fallback")
    # RJ_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    # (ranks share devices, the exchange is staged through the host); the default is RCCL.
    backend = os.environ.get("RJ_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    n = args.rows
    # RJ_BENCH_PROFILE=0 drops the per-kernel HIP events (diagnostic: what the bracketing costs)
    ctx = capi.Context(device=dev_index, profile=os.environ.get("RJ_BENCH_PROFILE", "1") != "0")
    rk, rp, sk, sp = make_relations(n, rank, world, device)
    if (args.zipf or args.payload64) and distributed:
        raise SystemExit("--zipf / --payload64 are single-GPU workloads")
    if args.zipf:
        g = torch.Generator(device=device)
        g.manual_seed(3)
        sk = zipf_keys(n, n, args.zipf, device, g)
    if args.payload64:
        rp, sp = rp.to(torch.int64) * 1_000_003, sp.to(torch.int64) * 1_000_003
    R = adopt(ctx, [rk, rp])
    S = adopt(ctx, [sk, sp])
    del rk, rp, sk, sp
    torch.cuda.empty_cache()

    if not distributed:
        plan = join_plan(pl.INT64 if args.payload64 else pl.INT32)

        def step():
            res = ctx.execute_resident(plan, [R, S])
            rows = res.num_rows
            res.free()
            return rows

    else:
        from pyrj import dist as rjdist

        sj = rjdist.ShardedJoin(rjdist.GpuOps(ctx, device))

        def step():
            res = sj.run(R, n, S, n)
            rows = res.num_rows
            res.free()
            return rows

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    rows = 0
    for _ in range(args.warmup):
        rows = step()
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        rows = step()
    barrier()
    dt = time.perf_counter() - t0
    stats = ctx.profile()

    total_rows = rows
    if distributed:
        rdev = device if backend == "nccl" else torch.device("cpu")
        tt = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        tr = torch.tensor([rows], dtype=torch.int64, device=rdev)
        dist.all_reduce(tr, op=dist.ReduceOp.SUM)
        total_rows = int(tr.item())
    # every probe key hits exactly one build row: |out| = |S|
    if total_rows != n * world:
        raise SystemExit(f"wrong result size: {total_rows} != {n * world}")

    if rank == 0:
        info = ctx.device_info()
        out = {
            "metric": "probe_tuples_per_sec",
            "value": n * world * args.steps / dt,
            "unit": "tuples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": f"single JoinNode, {n} x {n} INT32 keys per GPU (build = permutation, probe = "
                + (f"Zipf-{args.zipf}" if args.zipf else "uniform iid") + "), 1 "
                + ("INT64" if args.payload64 else "INT32") + " payload col per side, Page-packed inputs resident in HBM, Page-encoded output in HBM",
                "rows_per_relation_per_gpu": n,
                "parallelism": "single GPU" if world == 1 else f"hash-sharded x{world}, one all-to-all ({backend})",
                "device": info["name"],
                "arch": info["arch"],
            },
            "roofline": roofline(stats, n, n, args.steps, 8 if args.payload64 else 4) if stats else None,
        }
        if not distributed and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample)
            out["cpu_baseline"]["host_cores_available"] = os.cpu_count()
        print(json.dumps(out))
    R.release()
    S.release()
    ctx.destroy()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
