#!/usr/bin/env python3
"""End-to-end plan ms on JOB-shaped plans (BASELINE configs 1 and 5: job/1a.sql, job/13d.sql, ...).

The harness hands execute() already-filtered tables; IMDB and DuckDB are not available offline,
so every scan's input is a synthetic table with the IMDB schema whose row count is PostgreSQL's
own "Plan Rows" estimate for that scan (x3 parallel workers, capped by the real table size),
`id` a random subset of the real id range, foreign keys uniform over the parent's real id range.
Timed: Contest::execute semantics through the C-ABI with HOST pages in and out (rj_execute +
rj_result_copy_pages).  The oracle cross-check of these plans at this scale lives in
tests/test_gpu_job_plans.py::test_job_plans_at_scale.

usage: job_bench.py [--queries 1a,13d | all] [--repeat 3]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radix-join_amd")]
import numpy as np  # noqa: E402

from pyrj import capi, job, plan as pl  # noqa: E402

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", default="1a,13d")
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--max-input-rows", type=int, default=40_000_000)
    args = ap.parse_args()
    fx = job.load_fixture()
    names = sorted(fx["queries"]) if args.queries == "all" else args.queries.split(",")
    ctx = capi.build_context()
    rng = np.random.default_rng(7)
    cache = {}
    total_ms, rows_out = 0.0, 0
    for name in names:
        q = fx["queries"][name]
        tables = job.make_scaled_inputs(q, fx["schema"], rng, cache)
        plan = job.build_plan(q, fx["schema"], tables, by_alias=True)
        in_rows = sum(t.num_rows for t in plan.inputs)
        if in_rows > args.max_input_rows:
            print(f"{name}: skipped ({in_rows} input rows)")
            continue
        cplan, keep = pl.plan_to_c(plan)
        best = None
        for _ in range(args.repeat):
            t0 = time.perf_counter()
            out = C.c_void_p()
            ctx._check(ctx.L.rj_execute(ctx.h, C.byref(cplan), C.byref(out)))
            res = capi.Result(ctx, out)
            tbl = res.to_table()
            dt = (time.perf_counter() - t0) * 1e3
            res.free()
            best = dt if best is None else min(best, dt)
        line = f"{name}: joins={sum(1 for n in plan.nodes if isinstance(n.data, pl.JoinNode))} input_rows={in_rows} out_rows={tbl.num_rows} execute_ms={best:.1f}"
        print(line, flush=True)
        total_ms += best
        rows_out += tbl.num_rows
        del keep
    print(f"TOTAL {len(names)} queries: {total_ms:.0f} ms, {rows_out} output rows")
    capi.destroy_context(ctx)


if __name__ == "__main__":
    main()
