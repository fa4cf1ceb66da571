#!/usr/bin/env python3
"""SURVEY.md §8(f)#2, second half — would it pay to let a parent JoinNode skip its radix passes when
its input is a child join's result on the SAME key?  One survey over the 113 JOB plan trees (synthetic
IMDB-shaped inputs sized by PostgreSQL's Plan Rows, as in scripts/job_bench.py):
  * which joins are partitioned at all (build side > 4096 rows; the others take the broadcast join,
    which partitions nothing),
  * which partitioned joins read a partitioned child's result on that child's own key column (the
    only case where the child could emit in partition order and the parent skip its passes on that
    side),
  * what the partition kernels cost next to everything else (HIP-event totals, profile level 2).
usage (GPU box): python scripts/f2_survey.py > gpurun_out/f2_survey.log"""
import ctypes as C
import os
import re
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radix-join_amd")]
os.environ["RJ_DIAG"] = "2"
import numpy as np  # noqa: E402

from pyrj import capi, job, plan as pl  # noqa: E402

JN_RMAX = 4096


def post_order_joins(plan):
    out = []

    def walk(i):
        nd = plan.nodes[i]
        if isinstance(nd.data, pl.JoinNode):
            walk(nd.data.left)
            walk(nd.data.right)
            out.append(i)

    walk(plan.root)
    return out


def key_source(plan, i, side):
    """the child node and the index into ITS output list that join i's key on `side` reads"""
    j = plan.nodes[i].data
    return (j.left, j.left_attr) if side == 0 else (j.right, j.right_attr)


def is_childs_key(plan, child, out_idx):
    """does output column `out_idx` of join node `child` carry that join's key?"""
    nd = plan.nodes[child]
    if not isinstance(nd.data, pl.JoinNode):
        return False
    j = nd.data
    lw = len(plan.nodes[j.left].output_attrs)
    src = nd.output_attrs[out_idx][0]
    return src == j.left_attr or src == lw + j.right_attr


def main():
    fx = job.load_fixture()
    rng = np.random.default_rng(7)
    cache = {}
    ctx = capi.Context(profile=2)
    tot = {"joins": 0, "partitioned": 0, "reusable_sides": 0, "reusable_tuples": 0, "partitioned_tuples": 0}
    errfd = os.dup(2)
    for name in sorted(fx["queries"]):
        q = fx["queries"][name]
        tables = job.make_scaled_inputs(q, fx["schema"], rng, cache)
        plan = job.build_plan(q, fx["schema"], tables, by_alias=True)
        if sum(t.num_rows for t in plan.inputs) > 40_000_000:
            continue
        cplan, keep = pl.plan_to_c(plan)
        with tempfile.TemporaryFile() as tf:
            os.dup2(tf.fileno(), 2)
            try:
                h = C.c_void_p()
                ctx._check(ctx.L.rj_execute(ctx.h, C.byref(cplan), C.byref(h)))
                capi.Result(ctx, h).free()
            finally:
                os.dup2(errfd, 2)
            tf.seek(0)
            log = tf.read().decode(errors="replace")
        sizes = [(int(b), int(p)) for b, p in re.findall(r"\[rj diag\] join build=(\d+) probe=(\d+)", log)]
        joins = post_order_joins(plan)
        # joins with an empty child print nothing: only full traces are classified
        if len(sizes) != len(joins):
            continue
        part = {i: (b > JN_RMAX, b, p) for i, (b, p) in zip(joins, sizes)}
        for i in joins:
            tot["joins"] += 1
            is_part, b, p = part[i]
            if not is_part:
                continue
            tot["partitioned"] += 1
            tot["partitioned_tuples"] += b + p
            j = plan.nodes[i].data
            for side in (0, 1):
                child, idx = key_source(plan, i, side)
                if child in part and part[child][0] and is_childs_key(plan, child, idx):
                    tot["reusable_sides"] += 1
                    tot["reusable_tuples"] += b if (side == 0) == bool(j.build_left) else p
        del keep
    stats = ctx.profile() or []
    ms = {s["name"]: s["total_ms"] for s in stats}
    pass_ms = sum(v for k, v in ms.items() if k.startswith("pass") or k in ("scan_segments", "scan_fine", "group_table", "scan_bins"))
    all_ms = sum(ms.values())
    print("JOB plans, synthetic IMDB-shaped inputs:", tot)
    print("kernel ms over all plans: partition passes %.2f of %.2f (all kernels); by kernel:" % (pass_ms, all_ms),
          {k: round(v, 2) for k, v in sorted(ms.items(), key=lambda kv: -kv[1])[:12]})
    if tot["partitioned_tuples"]:
        print("share of partitioned tuples a partition-order emit could spare the parent: %.1f %%" %
              (100.0 * tot["reusable_tuples"] / tot["partitioned_tuples"]))
    ctx.destroy()


if __name__ == "__main__":
    main()
