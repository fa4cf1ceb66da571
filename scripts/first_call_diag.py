#!/usr/bin/env python3
"""Where does the first execute() of a fresh (prewarmed) context spend its time?  job/1a three times
with RJ_DIAG=2 (host-side timings on stderr) and wall clocks around every stage of the binding."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radix-join_amd")]
os.environ["RJ_DIAG"] = "2"
import numpy as np  # noqa: E402

from pyrj import capi, job, plan as pl  # noqa: E402

fx = job.load_fixture()
rng = np.random.default_rng(7)
q = fx["queries"][sys.argv[1] if len(sys.argv) > 1 else "1a"]
tables = job.make_scaled_inputs(q, fx["schema"], rng, {})
plan = job.build_plan(q, fx["schema"], tables, by_alias=True)
cplan, keep = pl.plan_to_c(plan)
t0 = time.perf_counter()
ctx = capi.Context(prewarm=True)
print(f"build_context (prewarm) {1e3 * (time.perf_counter() - t0):.2f} ms", file=sys.stderr)
for k in range(3):
    t0 = time.perf_counter()
    h = C.c_void_p()
    ctx._check(ctx.L.rj_execute(ctx.h, C.byref(cplan), C.byref(h)))
    t1 = time.perf_counter()
    res = capi.Result(ctx, h)
    tbl = res.to_table()
    t2 = time.perf_counter()
    res.free()
    print(f"call {k}: rj_execute {1e3 * (t1 - t0):.2f} ms, result copy-out {1e3 * (t2 - t1):.2f} ms, rows {tbl.num_rows}", file=sys.stderr)
ctx.destroy()
