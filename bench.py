#!/usr/bin/env python3
"""bench.py — probe-tuples/s of the radix join on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N = 1  headline workload = BASELINE.json configs[2] (the largest single-GPU configuration):
       single JoinNode, 1B ⋈ 1B INT32 keys, Zipf-0.9 probe keys, one INT64 payload column per
       side, plan Join(build_left=true, Scan(R){0,1}, Scan(S){0,1}, out={0,1,3}).
       One step = one rj_execute_resident() of that plan: Page-packed inputs already in HBM,
       page decode + histograms + 2 radix scatters per side + build/probe + Page-encoded result
       left in HBM, including the result-size read-back.  W untimed warmup steps, then exactly K
       timed steps.  After the timed loop the LAST result is verified on the device in closed
       form (every probe row exactly once, with its own key, next to the build row holding that
       key: pyrj/workloads.py::verify_pk_fk) — a wrong join fails the bench.
       The same JSON line carries, as extra fields, `configs` (BASELINE config 2 = 100M ⋈ 100M
       uniform, and 1B ⋈ 1B uniform, each timed and verified the same way) and `plan_ms`
       (end-to-end Contest::execute semantics, host pages in / host pages out, for the JOB plan
       trees job/1a, job/13d and job/10c — the last one a 1M-row result with two VARCHAR columns —
       over synthetic IMDB-shaped inputs: what the reference harness times,
       tests/read_sql.cpp:1234-1236).
N > 1  one process per GPU (torch.distributed, backend nccl = RCCL).  STRONG scaling: the same
       1B ⋈ 1B job, each rank holding 1/N of the rows of both relations; the join shards by key
       hash with one exchange step.  value = 1B probe tuples per
       max-over-ranks step time.

`roofline` = the kernel with the longest average launch (the build/probe kernel): algorithmic
bytes per launch over its HIP-event duration measured on the launch stream;
`roofline.kernels` lists the same for every hot kernel and `roofline.min_frac_kernel` names the
one furthest below the roofline.  `cpu_baseline` (N = 1) = the CPU oracle — a port of the
reference algorithm — on a bounded sample of the headline workload's shape.
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
from pyrj import workloads as wl  # noqa: E402

# kept importable from here (tests and scripts of round 1 use bench.adopt / bench.join_plan)
adopt, join_plan, pack_pages_gpu, pack_pages_gpu64, zipf_keys = wl.adopt, wl.join_plan, wl.pack_pages_gpu, wl.pack_pages_gpu64, wl.zipf_keys

HBM_PEAK_GBPS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8 TB/s; ~6.3 TB/s achievable)
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic_r03.json")


def algo_bytes(payload_bytes):
    """Algorithmic bytes per tuple of each kernel (k = 4-byte key, p = carry bytes):
    a radix pass costs k for its histogram read and (k+p) read + (k+p) write for its scatter; the
    probe phase costs (k+pS) + (k+pR) + output row (k + pR + pS) per probe tuple."""
    k, p = 4.0, float(payload_bytes)
    return {
        "pass1_hist": k,
        "pass1_scatter": 2 * (k + p),
        "pass2_hist": k,
        "pass2_scatter": 2 * (k + p),
        "join_build_probe": (k + p) + (k + p) + (k + 2 * p),
    }


def roofline(stats, rows_build, rows_probe, steps_profiled, payload_bytes, workload):
    ab = algo_bytes(payload_bytes)
    traffic_all = {}
    try:
        with open(TRAFFIC_FILE) as f:
            traffic_all = json.load(f).get(workload, {})
    except Exception:
        traffic_all = {}
    kernels = {}
    for s in stats:
        name = s["name"]
        if name not in ab or not s["launches"]:
            continue
        avg_ms = s["total_ms"] / s["launches"]
        # partition kernels run once per relation per step (equal cardinalities here)
        tuples = rows_probe if name == "join_build_probe" else (rows_build + rows_probe) / 2.0
        algo = ab[name] * tuples
        ach = algo / (avg_ms * 1e-3) / 1e9
        kernels[name] = {
            "achieved": ach,
            "frac": ach / HBM_PEAK_GBPS,
            "avg_launch_ms": avg_ms,
            "launches_per_step": s["launches"] / steps_profiled,
            "ms_per_step": s["total_ms"] / steps_profiled,
            "algorithmic_bytes_per_launch": algo,
            "traffic": traffic_all.get("bytes_per_launch", {}).get(name),
        }
    if not kernels:
        return None
    # dominant = longest average launch (ties by name: deterministic)
    dom = max(sorted(kernels), key=lambda n: kernels[n]["avg_launch_ms"])
    d = kernels[dom]
    # the north-star bar as SURVEY.md §8(d) defines it: the probe phase's READ bytes (partitioned S
    # tuple + partitioned R tuple per probe tuple) over the build/probe kernel's time, against the
    # 8 TB/s spec — the kernel also writes the materialised output in that time, so this is below
    # `frac` (all bytes) by construction
    probe_read = None
    if "join_build_probe" in kernels:
        j = kernels["join_build_probe"]
        rd = 2 * (4.0 + payload_bytes) * rows_probe
        probe_read = {"read_bytes_per_launch": rd, "achieved": rd / (j["avg_launch_ms"] * 1e-3) / 1e9,
                      "frac": rd / (j["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    return {
        "kernel": dom,
        "bound": "hbm",
        "achieved": d["achieved"],
        "peak": HBM_PEAK_GBPS,
        "unit": "GB/s",
        "frac": d["frac"],
        "traffic": d["traffic"],
        "traffic_source": (traffic_all.get("source") if d["traffic"] else None),
        # `traffic` replays the PMC passes committed under profiles/ (counters cannot be collected
        # inside a timed run); `achieved` and the kernel times ARE measured in this run
        "traffic_measured_in_run": False,
        "probe_read_frac": probe_read["frac"] if probe_read else None,
        "probe_read": probe_read,
        "avg_launch_ms": d["avg_launch_ms"],
        "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"],
        "min_frac_kernel": min(sorted(kernels), key=lambda n: kernels[n]["frac"]),
        "kernels": kernels,
    }


def run_single(name, device, dev_index, steps, warmup, verify=True, rows=None):
    """One workload on one GPU: build it on the device, W warmup + K timed steps, verify the last
    result in closed form.  Own context, destroyed at the end (releases its HBM cache)."""
    ctx = capi.Context(device=dev_index, profile=os.environ.get("RJ_BENCH_PROFILE", "1") != "0")
    rel = wl.make_relations(name, device, rows=rows)
    R = wl.adopt(ctx, [rel.rk, rel.rp])
    S = wl.adopt(ctx, [rel.sk, rel.sp])
    torch.cuda.empty_cache()
    plan = wl.join_plan(rel.payload_type)
    n = rel.n

    def step(keep=False):
        res = ctx.execute_resident(plan, [R, S])
        if keep:
            return res
        rows_out = res.num_rows
        res.free()
        return rows_out

    for _ in range(warmup):
        step()
    ctx.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for i in range(steps):
        if i + 1 == steps:
            last = step(keep=True)
        else:
            step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = ctx.profile()
    if last.num_rows != n:
        raise SystemExit(f"{name}: wrong result size {last.num_rows} != {n}")
    digest = None
    if verify:
        try:
            digest = wl.verify_pk_fk(last, rel)
        except AssertionError as e:
            raise SystemExit(f"{name}: RESULT VERIFICATION FAILED: {e}")
    last.free()
    info = ctx.device_info()
    out = {
        "workload": name,
        "label": wl.WORKLOADS[name]["label"],
        "rows_per_relation": n,
        "value": n * steps / dt,
        "ms_per_step": dt / steps * 1e3,
        "steps": steps,
        "warmup": warmup,
        "verified": digest,
        "roofline": roofline(stats, n, n, steps, 8 if rel.payload64 else 4, name) if stats else None,
    }
    R.release()
    S.release()
    ctx.destroy()
    del rel, R, S
    torch.cuda.empty_cache()
    return out, info


def job_plan_ms(dev_index, queries=("1a", "13d", "10c"), repeat=3):
    """End-to-end plan ms, host pages in / host pages out (rj_execute + rj_result_copy_pages),
    on JOB plan trees over synthetic IMDB-shaped inputs (IMDB itself is not available offline):
    what the reference harness times around Contest::execute (tests/read_sql.cpp:1234-1236)."""
    import ctypes as C

    from pyrj import job

    fx = job.load_fixture()
    rng = np.random.default_rng(7)
    cache = {}
    # what Contest::build_context() does (RJ_CTX_PREWARM): the one-off set-up costs land in the
    # context's construction, which the harness times once (tests/read_sql.cpp:1279-1283), not in
    # the first query
    t0 = time.perf_counter()
    ctx = capi.Context(device=dev_index, prewarm=True)
    build_context_ms = (time.perf_counter() - t0) * 1e3
    out = {"build_context_ms": build_context_ms}
    for name in queries:
        q = fx["queries"][name]
        tables = job.make_scaled_inputs(q, fx["schema"], rng, cache)
        plan = job.build_plan(q, fx["schema"], tables, by_alias=True)
        cplan, keep = pl.plan_to_c(plan)
        times = []
        rows_out = 0
        for _ in range(repeat + 1):  # first run warms the HBM block cache and pinned staging
            t0 = time.perf_counter()
            h = C.c_void_p()
            ctx._check(ctx.L.rj_execute(ctx.h, C.byref(cplan), C.byref(h)))
            res = capi.Result(ctx, h)
            tbl = res.to_table()
            times.append((time.perf_counter() - t0) * 1e3)
            rows_out = tbl.num_rows
            res.free()
        del keep
        # the same plan over RESIDENT inputs (what a harness that ingests on the device gets,
        # rj_table_from_csv -> rj_execute_resident): no PCIe upload inside execute(), result pages
        # still fetched to the host
        resident_ms = None
        try:
            up = [ctx.upload(t) for t in plan.inputs]
            rt = []
            for _ in range(repeat + 1):
                t0 = time.perf_counter()
                r = ctx.execute_resident(plan, up, keep_on_device=False)
                tbl2 = r.to_table()
                rt.append((time.perf_counter() - t0) * 1e3)
                r.free()
            assert tbl2.num_rows == rows_out
            resident_ms = min(rt[1:])
            for u in up:
                u.release()
        except Exception as e:  # noqa: BLE001
            resident_ms = f"{type(e).__name__}: {e}"[:200]
        out[f"job/{name}"] = {
            "ms": min(times[1:]),
            "ms_resident_inputs": resident_ms,
            "ms_first_call": times[0],
            "joins": sum(1 for nd in plan.nodes if isinstance(nd.data, pl.JoinNode)),
            "input_rows": int(sum(t.num_rows for t in plan.inputs)),
            "output_rows": int(rows_out),
            "data": "synthetic IMDB-shaped (row counts = PostgreSQL Plan Rows of plans.json)",
            "reference_published_ms": {"1a": 3315, "13d": 11770, "10c": 5228}.get(name),  # benchmarks/run_b78733e.txt:1,48,37 (TR PRO 7995WX, real IMDB)
        }
    ctx.destroy()
    return out


def ingest_ms(dev_index, rows=1_000_000, repeat=3, with_cpu=True):
    """Table::from_csv on the device (rj_table_from_csv; SURVEY.md §8f-4 — runs before execute() in
    the harness and is untimed there): a cast_info-shaped CSV (INT32 id, nullable INT32, quoted
    VARCHAR with a comma, INT32) of `rows` records in host memory -> parsed, filtered (two numeric
    predicates) and Page-packed in HBM.  Timed: the whole call incl. the upload of the text; beside
    it the CPU oracle's restatement of the reference path on the same text (one thread)."""
    rng = np.random.default_rng(11)
    a, b, d = rng.integers(0, 4_000_000, rows), rng.integers(0, 1000, rows), rng.integers(1, 12, rows)
    text = b"".join(b"%d,%s,\"(as %d, uncredited)\",%d\n" % (a[i], b"" if b[i] < 70 else b"%d" % b[i], b[i], d[i]) for i in range(rows))
    types = [pl.INT32, pl.INT32, pl.VARCHAR, pl.INT32]
    filt = [("LT", 3, 5), ("IS_NOT_NULL", 1), ("AND",)]
    ctx = capi.Context(device=dev_index, prewarm=True)
    times, kept = [], 0
    for _ in range(repeat + 1):
        t0 = time.perf_counter()
        t = ctx.from_csv(text, types, filt)
        times.append((time.perf_counter() - t0) * 1e3)
        kept = int(ctx.L.rj_table_num_rows(t.h)) if hasattr(ctx.L, "rj_table_num_rows") else 0
        t.release()
    ctx.destroy()
    want_rows = int(((d < 5) & (b >= 70)).sum())  # closed form of the filter
    if want_rows != kept:
        raise SystemExit(f"ingest: the device kept {kept} rows, expected {want_rows}")
    cpu_ms = None
    if with_cpu:  # the CPU oracle's restatement of the reference path, as a reported baseline (one thread)
        import _oracle

        t0 = time.perf_counter()
        want = _oracle.from_csv(text, types, filt)
        cpu_ms = (time.perf_counter() - t0) * 1e3
        if want.num_rows != kept:
            raise SystemExit(f"ingest: the device kept {kept} rows, the oracle {want.num_rows}")
    best = min(times[1:])
    return {"ms": best, "ms_first_call": times[0], "text_mb": len(text) / 1e6, "rows": rows, "rows_kept": kept,
            "text_gb_per_s": len(text) / best / 1e6, "cpu_oracle_ms": cpu_ms, "cpu_cores": 1,
            "what": "rj_table_from_csv: CSV text in host memory -> filtered resident table (upload + parse + filter + page fill)"}


def _config3_sample(n, seed=1):
    """unique INT32 build keys, Zipf-0.9 probe keys over them, INT64 payloads (numpy)"""
    rng = np.random.default_rng(seed)
    rk = rng.permutation(n).astype(np.int32)
    w = 1.0 / np.arange(1, n + 1, dtype=np.float64) ** 0.9
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    ranks = np.searchsorted(cdf, rng.random(n), side="right").clip(max=n - 1).astype(np.int64)
    del w, cdf
    sk = ((ranks * 7919 + 13) % n).astype(np.int32)
    return rk, sk


_CPU_SHARDS = None  # inherited by the forked workers (no pickling of the arrays)


def _cpu_worker(args):
    shard, barrier = args
    import _oracle

    rk, rp, sk, sp = _CPU_SHARDS[shard]
    p = wl.join_plan(pl.INT64)
    p.new_input(pl.make_table([(pl.INT32, rk), (pl.INT64, rp)]))
    p.new_input(pl.make_table([(pl.INT32, sk), (pl.INT64, sp)]))
    _oracle.lib()
    barrier.wait()  # every worker starts its join at the same moment
    t0 = time.perf_counter()
    res = _oracle.execute(p)
    dt = time.perf_counter() - t0
    assert res.num_rows == sk.shape[0]
    return sk.shape[0], dt


def cpu_baseline(sample_rows, workers=0):
    """The CPU oracle (port of the reference's execute path, oracle/rjo_oracle.c) on a bounded
    sample of the headline workload's shape (unique INT32 build keys, Zipf-0.9 probe keys, INT64
    payloads), on this box's host cores.  The port is single-threaded, as is the fastest
    configuration of the reference itself (its 8-thread run was slower than 1 thread, SURVEY.md
    §6); the all-core figure runs one instance per core on hash-disjoint shards of the same
    sample (key mod P: PK-FK shards join independently), all started together.
    Must run before this process touches the GPU (the workers are forked)."""
    global _CPU_SHARDS
    import multiprocessing as mp

    import _oracle

    n = sample_rows
    rk, sk = _config3_sample(n)
    pay = np.arange(n, dtype=np.int64) * wl.PAY_MUL
    # ---- one thread, the whole sample
    p = wl.join_plan(pl.INT64)
    p.new_input(pl.make_table([(pl.INT32, rk), (pl.INT64, pay)]))
    p.new_input(pl.make_table([(pl.INT32, sk), (pl.INT64, pay)]))
    _oracle.lib()
    t0 = time.perf_counter()
    res = _oracle.execute(p)
    dt1 = time.perf_counter() - t0
    assert res.num_rows == n
    del res, p
    # ---- P instances on hash-disjoint shards (16 = the box's CPU share for one GPU)
    P = workers or min(16, os.cpu_count() or 1)
    _CPU_SHARDS = []
    for s in range(P):
        mb, ms = (rk % P) == s, (sk % P) == s
        _CPU_SHARDS.append((rk[mb], pay[mb], sk[ms], pay[ms]))
    ctx = mp.get_context("fork")
    with mp.Manager() as mgr:
        barrier = mgr.Barrier(P)
        with ctx.Pool(P) as pool:
            outs = pool.map(_cpu_worker, [(s, barrier) for s in range(P)])
    _CPU_SHARDS = None
    rows = sum(o[0] for o in outs)
    wall = max(o[1] for o in outs)
    return {
        "value": rows / wall,
        "unit": "probe tuples/s",
        "cores": P,
        "kind": "port",
        "sample": f"{n} x {n} INT32 keys (Zipf-0.9 probe), INT64 payloads, same plan, oracle/rjo_oracle.c incl. page decode+encode: "
        f"{P} single-threaded instances on key-mod-{P} shards started together, slowest {wall:.1f} s",
        "single_thread": {"value": n / dt1, "cores": 1, "seconds": dt1, "sample": f"the whole {n} x {n} sample in one instance"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="config3", choices=sorted(wl.WORKLOADS), help="headline workload at N=1")
    ap.add_argument("--rows", type=int, default=0, help="rows per relation overall (default: the workload's own size)")
    ap.add_argument("--cpu-sample", type=int, default=40_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra `configs` and `plan_ms` fields")
    ap.add_argument("--no-verify", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # RJ_BENCH_FORCE_DIST=1: run the sharded path at world size 1 (rehearsal on a one-GPU box,
    # under `python -m torch.distributed.run --nproc-per-node 1`)
    distributed = world > 1 or os.environ.get("RJ_BENCH_FORCE_DIST") == "1"
    if args.gpus != world and (distributed or args.gpus > 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N > 1 as `python -m torch.distributed.run "
                         f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...`")
    cpu = None
    if not distributed and not args.no_cpu_baseline:
        # first, while this process has not touched the GPU yet (the workers are forked); it is
        # a reported baseline on the host cores, not part of any timed GPU region
        if torch.cuda.device_count() == 0:
            raise SystemExit("bench.py needs a GPU: the HIP path is the product, there is no CPU fallback")
        try:
            cpu = cpu_baseline(args.cpu_sample)
            cpu["host_cores_available"] = os.cpu_count()
        except Exception as e:  # noqa: BLE001  (the GPU line is still worth having)
            cpu = {"value": None, "unit": "probe tuples/s", "cores": 0, "kind": "port", "sample": "failed",
                   "error": f"{type(e).__name__}: {e}"[:300]}
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path is the product, there is no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    if distributed:
        import bench_dist

        return bench_dist.main(args, rank, world, dev_index, device)

    head, info = run_single(args.workload, device, dev_index, args.steps, args.warmup, verify=not args.no_verify, rows=args.rows or None)
    out = {
        "metric": "probe_tuples_per_sec",
        "value": head["value"],
        "unit": "tuples/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"],
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {
            "workload": head["label"] + "; Page-packed inputs resident in HBM, Page-encoded output left in HBM",
            "rows_per_relation": head["rows_per_relation"],
            "parallelism": "single GPU",
            "device": info["name"],
            "arch": info["arch"],
            "verified": head["verified"],
        },
        "roofline": head["roofline"],
    }
    if not args.no_extras:
        # the extras never take the headline line with them: a failure is recorded in its place
        def guarded(fn, *a, **kw):
            try:
                return fn(*a, **kw)
            except Exception as e:  # noqa: BLE001
                return {"error": f"{type(e).__name__}: {e}"[:300]}

        extras = {}
        for name, st, wu in (("config2", 10, 2), ("uniform1b", 5, 1)):
            if name == args.workload:
                continue
            got = guarded(run_single, name, device, dev_index, st, wu, verify=not args.no_verify)
            if isinstance(got, dict):
                extras[name] = got
                continue
            r, _ = got
            rf = r["roofline"] or {}
            extras[name] = {
                "label": r["label"],
                "value": r["value"],
                "unit": "tuples/s",
                "ms_per_step": r["ms_per_step"],
                "steps": st,
                "warmup": wu,
                "verified": r["verified"],
                "kernels_ms_per_step": {k: v["ms_per_step"] for k, v in (rf.get("kernels") or {}).items()},
                "kernels_frac": {k: v["frac"] for k, v in (rf.get("kernels") or {}).items()},
            }
        out["configs"] = extras
        out["plan_ms"] = guarded(job_plan_ms, dev_index)
        out["ingest"] = guarded(ingest_ms, dev_index, with_cpu=not args.no_cpu_baseline)
    if cpu is not None:
        out["cpu_baseline"] = cpu
    print(json.dumps(out))


if __name__ == "__main__":
    main()
