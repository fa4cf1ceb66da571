#!/usr/bin/env python3
"""No-cliff check across join shapes the hot path specialises for: one JoinNode, N x N rows (unique
build keys, every probe key hits once unless stated), inputs resident in HBM, result left in HBM.
Prints G probe tuples/s and ms per join for each shape — key type, payload columns per side, NULLs,
duplicates — so that a shape that falls off the fast paths (row-index carries + gather, 64-bit
keys, three- and four-word tuples) shows next to the BASELINE shapes.
    python scripts/shape_sweep.py [rows]        (default 100 M; on the GPU box)
RJ_SWEEP_PROFILE=1 adds the per-kernel milliseconds of one join per shape (timed runs then carry the
profiler's events)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "radix-join_amd"))
from pyrj import capi  # noqa: E402
from pyrj import plan as pl  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
I32, I64, F64 = pl.INT32, pl.INT64, pl.FP64
NP = {I32: np.int32, I64: np.int64, F64: np.float64}
PROFILE = os.environ.get("RJ_SWEEP_PROFILE") == "1"
rng = np.random.default_rng(1)


def keycol(kt, base):
    if kt == I32:
        return base.astype(np.int32)
    if kt == I64:
        return base.astype(np.int64) * 4_000_000_007 - 12345
    return base.astype(np.float64) * 0.5 - 1000.0


def paycol(dt, n, null=0.0):
    v = rng.integers(-(2**30), 2**30, n).astype(NP[dt])
    return (dt, v, rng.random(n) >= null) if null else (dt, v)


SHAPES = [
    # name, key type, build payloads, probe payloads, build-key duplication, NULL fraction of the first build payload
    ("k32 | 32 / 32   (config 2)", I32, [I32], [I32], 1, 0.0),
    ("k32 | 64 / 64   (config 3)", I32, [I64], [I64], 1, 0.0),
    ("k32 | 32 / 64", I32, [I32], [I64], 1, 0.0),
    ("k64 | 32 / 32", I64, [I32], [I32], 1, 0.0),
    ("k64 | 64 / 64", I64, [I64], [I64], 1, 0.0),
    ("f64 | 32 / 32", F64, [I32], [I32], 1, 0.0),
    ("k32 | 32+32 / 32   (wide, 2 words)", I32, [I32, I32], [I32], 1, 0.0),
    ("k32 | 32+64 / 32   (wide, 3 words)", I32, [I32, I64], [I32], 1, 0.0),
    ("k32 | 32 (10% NULL) / 32   (wide: value + validity)", I32, [I32], [I32], 1, 0.1),
    ("k32 | 32+64+32 / 32   (row index + gather)", I32, [I32, I64, I32], [I32], 1, 0.0),
    ("k32 | 32 / 32, every build key twice", I32, [I32], [I32], 2, 0.0),
    ("k32 | - / -   (keys only)", I32, [], [], 1, 0.0),
    ("k32 | 64 / 32   (the tuple layouts of the 2-word wide shape, one column)", I32, [I64], [I32], 1, 0.0),
]


def run(ctx, name, kt, bpay, ppay, dup, null):
    nb = N
    dom = nb // dup
    bbase = rng.permutation(dom)
    if dup > 1:
        bbase = np.concatenate([bbase] * dup)
    pbase = rng.integers(0, dom, N)
    bcols = [(kt, keycol(kt, bbase))] + [paycol(dt, nb, null if i == 0 else 0.0) for i, dt in enumerate(bpay)]
    pcols = [(kt, keycol(kt, pbase))] + [paycol(dt, N) for dt in ppay]
    bt, pt = pl.make_table(bcols), pl.make_table(pcols)
    p = pl.Plan()
    b = p.new_scan_node(0, [(i, c[0]) for i, c in enumerate(bcols)])
    s = p.new_scan_node(1, [(i, c[0]) for i, c in enumerate(pcols)])
    both = [c[0] for c in bcols] + [c[0] for c in pcols]
    out = [(i, t) for i, t in enumerate(both) if i != len(bcols)]  # the probe side's key column once is enough
    p.root = p.new_join_node(True, b, s, 0, 0, out)
    p.new_input(bt)
    p.new_input(pt)
    B, S = ctx.upload(bt), ctx.upload(pt)
    del bt, pt, bcols, pcols
    rows = None
    for _ in range(2):
        r = ctx.execute_resident(p, [B, S])
        rows = r.num_rows
        r.free()
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        ctx.execute_resident(p, [B, S]).free()
    ms = (time.perf_counter() - t0) / K * 1e3
    assert rows == N * dup, (rows, N * dup)
    print("%-58s %8.2f ms  %6.2f G probe tuples/s" % (name, ms, N / ms / 1e6), flush=True)
    if PROFILE:  # per-kernel ms of one more join (the library brackets every launch)
        ctx.profile_reset()
        ctx.execute_resident(p, [B, S]).free()
        print("      " + "  ".join("%s=%.2f" % (k["name"], k["total_ms"]) for k in ctx.profile() if k["total_ms"] >= 0.005), flush=True)
    B.release()
    S.release()


def run_tree(ctx, name, n, wide):
    """(A join B) join C, n rows each: B.k -> A.k (every B row finds its A row), then the result's
    B.fk -> C.k.  `wide`: the first join hands two payload columns per side to the second."""
    a_k = rng.permutation(n).astype(np.int32)
    c_k = rng.permutation(n).astype(np.int32)
    acols = [(I32, a_k), paycol(I32, n)] + ([paycol(I64, n)] if wide else [])
    bcols = [(I32, rng.integers(0, n, n).astype(np.int32)), (I32, rng.integers(0, n, n).astype(np.int32)), paycol(I32, n)]
    ccols = [(I32, c_k), paycol(I64, n)]
    ta, tb, tc = pl.make_table(acols), pl.make_table(bcols), pl.make_table(ccols)
    p = pl.Plan()
    sa = p.new_scan_node(0, [(i, c[0]) for i, c in enumerate(acols)])
    sb = p.new_scan_node(1, [(i, c[0]) for i, c in enumerate(bcols)])
    na = len(acols)
    # out of the first join: A's payloads, B.fk, B.y
    out1 = [(i, acols[i][0]) for i in range(1, na)] + [(na + 1, I32), (na + 2, I32)]
    j1 = p.new_join_node(True, sa, sb, 0, 0, out1)
    sc = p.new_scan_node(2, [(0, I32), (1, I64)])
    n1 = len(out1)
    fk_at = n1 - 2
    out2 = [(i, out1[i][1]) for i in range(n1)] + [(n1 + 1, I64)]
    p.root = p.new_join_node(False, j1, sc, fk_at, 0, out2)  # probe = the intermediate, build = C
    for t in (ta, tb, tc):
        p.new_input(t)
    T = [ctx.upload(t) for t in (ta, tb, tc)]
    del ta, tb, tc
    rows = None
    for _ in range(2):
        r = ctx.execute_resident(p, T)
        rows = r.num_rows
        r.free()
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        ctx.execute_resident(p, T).free()
    ms = (time.perf_counter() - t0) / K * 1e3
    assert rows == n, (rows, n)
    print("%-58s %8.2f ms  %6.2f G rows/s through two joins" % (name, ms, n / ms / 1e6), flush=True)
    if PROFILE:
        ctx.profile_reset()
        ctx.execute_resident(p, T).free()
        print("      " + "  ".join("%s=%.2f" % (k["name"], k["total_ms"]) for k in ctx.profile() if k["total_ms"] >= 0.005), flush=True)
    for t in T:
        t.release()


def main():
    ctx = capi.Context(profile=2 if PROFILE else False)  # 2: every kernel, not only the hot five
    print(f"{N} x {N} rows, inputs resident, result left in HBM; 5 timed joins after 2 warm-ups")
    only = os.environ.get("RJ_SWEEP_ONLY")  # e.g. "0,6": shapes by index
    for i, sh in enumerate(SHAPES):
        if only is None or str(i) in only.split(","):
            run(ctx, *sh)
    if only is None or "tree" in only.split(","):
        run_tree(ctx, "(A x B) x C, one payload column per table", N // 2, False)
        run_tree(ctx, "(A x B) x C, A carries INT32 + INT64 (wide)", N // 2, True)
    capi.destroy_context(ctx)


if __name__ == "__main__":
    main()
