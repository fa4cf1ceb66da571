#!/usr/bin/env python3
"""End-to-end plan ms on JOB-shaped plans (BASELINE configs 1 and 5: job/1a.sql, job/13d.sql, ...).

The harness hands execute() already-filtered tables; IMDB and DuckDB are not available offline,
so every scan's input is a synthetic table with the IMDB schema whose row count is PostgreSQL's
own "Plan Rows" estimate for that scan (x3 parallel workers, capped by the real table size),
`id` a random subset of the real id range, foreign keys uniform over the parent's real id range.
Timed: Contest::execute semantics through the C-ABI with HOST pages in and out (rj_execute +
rj_result_copy_pages).  The same Plan goes through the CPU oracle for the row count (and its
time, as the CPU figure on this box).

usage: job_bench.py [--queries 1a,13d | all] [--repeat 3] [--oracle]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radix-join_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402

from pyrj import capi, job, pages as pg, plan as pl  # noqa: E402

REAL_ROWS = {  # IMDB (JOB snapshot) cardinalities
    "aka_name": 901343, "aka_title": 361472, "cast_info": 36244344, "char_name": 3140339, "comp_cast_type": 4,
    "company_name": 234997, "company_type": 4, "complete_cast": 135086, "info_type": 113, "keyword": 134170,
    "kind_type": 7, "link_type": 18, "movie_companies": 2609129, "movie_info": 14835720, "movie_info_idx": 1380035,
    "movie_keyword": 4523930, "movie_link": 29997, "name": 4167491, "person_info": 2963664, "role_type": 12,
    "title": 2528312,
}


def make_scan_table(schema_cols, table, n, rng):
    t = pl.ColumnarTable(n, [])
    for name, typ, nullable in schema_cols:
        if typ == "INT32":
            if name == "id":
                vals = rng.choice(REAL_ROWS[table], size=n, replace=False).astype(np.int32) if n < REAL_ROWS[table] else rng.permutation(n).astype(np.int32)
            elif name in job.FK_PARENT:
                vals = rng.integers(0, REAL_ROWS[job.FK_PARENT[name]], n).astype(np.int32)
            else:
                vals = rng.integers(1880, 2025, n).astype(np.int32)
            valid = (rng.random(n) >= 0.05) if nullable else None
            t.columns.append(pl.Column(pl.INT32, pg.pack_fixed(vals, valid, pl.INT32)))
        else:
            t.columns.append(pl.Column(pl.VARCHAR, pg.pack_varchar_fixed(np.arange(n), digits=9, prefix=name[:3].encode())))
    return t


def build(query, schema, rng, cache):
    tables = {}

    def visit(tree):
        if "scan" in tree:
            n = max(1, min(REAL_ROWS[tree["scan"]], tree["rows"] * 3 if tree["filtered"] else REAL_ROWS[tree["scan"]]))
            key = (tree["scan"], n)
            if key not in cache:
                cache[key] = make_scan_table(schema[tree["scan"]], tree["scan"], n, rng)
            tables[tree["alias"]] = cache[key]
        else:
            visit(tree["left"])
            visit(tree["right"])

    visit(query["tree"])
    return tables


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", default="1a,13d")
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--oracle", action="store_true")
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
        tables = build(q, fx["schema"], rng, cache)
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
        if args.oracle:
            import _oracle

            t0 = time.perf_counter()
            want = _oracle.execute(plan)
            line += f" oracle_ms={(time.perf_counter() - t0) * 1e3:.0f} rows_match={want.num_rows == tbl.num_rows}"
            assert want.num_rows == tbl.num_rows
        print(line, flush=True)
        total_ms += best
        rows_out += tbl.num_rows
        del keep
    print(f"TOTAL {len(names)} queries: {total_ms:.0f} ms, {rows_out} output rows")
    capi.destroy_context(ctx)


if __name__ == "__main__":
    main()
