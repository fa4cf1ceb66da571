#!/usr/bin/env python3
"""End-to-end Contest::execute timing with HOST pages in and out (PCIe inclusive):
upload (gather into pinned staging + H2D) + kernels + result D2H into caller pages.
usage: e2e_host.py [rows]   (BASELINE config 2 shape; default 100M x 100M)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radix-join_amd")]
import numpy as np  # noqa: E402

from pyrj import capi, plan as pl  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
rng = np.random.default_rng(1)
rk = rng.permutation(n).astype(np.int32)
sk = rng.integers(0, n, n).astype(np.int32)
pay = np.arange(n, dtype=np.int32)
p = pl.Plan()
p.new_scan_node(0, [(0, pl.INT32), (1, pl.INT32)])
p.new_scan_node(1, [(0, pl.INT32), (1, pl.INT32)])
p.new_join_node(True, 0, 1, 0, 0, [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)])
p.new_input(pl.make_table([(pl.INT32, rk), (pl.INT32, pay)]))
p.new_input(pl.make_table([(pl.INT32, sk), (pl.INT32, pay)]))
p.root = 2
ctx = capi.build_context()
import ctypes as C  # noqa: E402

cplan, keep = pl.plan_to_c(p)
for it in range(3):
    t0 = time.perf_counter()
    out = C.c_void_p()
    ctx._check(ctx.L.rj_execute(ctx.h, C.byref(cplan), C.byref(out)))
    t1 = time.perf_counter()
    res = capi.Result(ctx, out)
    rows = res.num_rows
    # result pages -> caller-owned host pages (what the shim does into `new Page`s)
    cols, t_alloc = [], 0.0
    for c in range(res.num_cols):
        npg = res.col_pages(c)
        ta = time.perf_counter()
        pages = np.empty((npg, 8192), dtype=np.uint8)  # untouched: first-touch faults land in the copy
        ptrs = pages.ctypes.data + np.arange(npg, dtype=np.uint64) * np.uint64(8192)
        t_alloc += time.perf_counter() - ta
        ctx._check(ctx.L.rj_result_copy_pages(res.h, c, ptrs.ctypes.data_as(C.POINTER(C.c_void_p)), npg))
        cols.append((pages, ptrs))
    t2 = time.perf_counter()
    # the same copy again into the now-resident pages: what the first-touch faults cost
    for c, (pages, ptrs) in enumerate(cols):
        ctx._check(ctx.L.rj_result_copy_pages(res.h, c, ptrs.ctypes.data_as(C.POINTER(C.c_void_p)), len(ptrs)))
    t3 = time.perf_counter()
    res.free()
    in_gb = 4 * p.inputs[0].columns[0].pages.nbytes / 1e9
    out_gb = sum(c[0].nbytes for c in cols) / 1e9
    print(f"run {it}: rows={rows} execute(upload+kernels)={1e3*(t1-t0):.1f} ms  copy_pages={1e3*(t2-t1):.1f} ms  "
          f"(again, pages resident: {1e3*(t3-t2):.1f} ms)  total={1e3*(t2-t0):.1f} ms  in={in_gb:.2f} GB out={out_gb:.2f} GB  -> {n/(t2-t0)/1e9:.2f} G probe tuples/s end to end")
assert rows == n
capi.destroy_context(ctx)
