#!/usr/bin/env python3
"""Condense the reference's JOB workload description into a small fixture.

Reads DATA files of the reference (run where /root/reference exists):
    plans.json          PostgreSQL EXPLAIN (FORMAT JSON) trees of the 113 JOB queries
    job/<name>.sql      the queries (SELECT list = MIN(alias.col) ..., FROM table AS alias, ...)
    job/schema.sql      IMDB schema (column names, integer vs text, NOT NULL)
and writes tests/golden/job_plans.json: per query the join tree (scan = table+alias, hash join =
condition + which child is the build side) and the output columns; plus the schema.  This is
what the reference's harness turns into a `Plan` (tests/read_sql.cpp:861-1141, rules in
SURVEY.md Appendix B); filters are dropped (they run before execute(), on the host).
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "job_plans.json")


def schema():
    txt = open(os.path.join(REF, "job", "schema.sql")).read()
    tables = {}
    for m in re.finditer(r"CREATE TABLE (\w+) \((.*?)\);", txt, re.S):
        cols = []
        for line in m.group(2).split("\n"):
            line = line.strip().rstrip(",")
            if not line:
                continue
            name, rest = line.split(None, 1)
            typ = "INT32" if rest.startswith("integer") else "VARCHAR"
            cols.append([name, typ, 0 if "NOT NULL" in rest else 1])
        tables[m.group(1)] = cols
    return tables


def strip(node):
    while node["Node Type"] in ("Aggregate", "Gather", "Gather Merge"):
        node = node["Plans"][0]
    return node


def aliases(t):
    return {t["alias"]} if "scan" in t else aliases(t["left"]) | aliases(t["right"])


def tree(node):
    node = strip(node)
    nt = node["Node Type"]
    if nt in ("Seq Scan", "Index Only Scan", "Index Scan", "Bitmap Heap Scan"):
        # "Plan Rows" = PostgreSQL's estimate of the scan's output after its filter, per parallel
        # worker: sizes the synthetic input of this scan in scripts/job_bench.py
        return {"scan": node["Relation Name"], "alias": node["Alias"], "rows": int(node.get("Plan Rows", 0)),
                "filtered": "Filter" in node or "Index Cond" in node}
    if nt != "Hash Join":
        raise ValueError(nt)
    assert node.get("Join Type", "Inner") == "Inner"
    a, b = node["Plans"]
    build_left = a["Node Type"] == "Hash"  # reference read_sql.cpp:943-953
    left = tree(a["Plans"][0] if a["Node Type"] == "Hash" else a)
    right = tree(b["Plans"][0] if b["Node Type"] == "Hash" else b)
    m = re.fullmatch(r"\((\w+)\.(\w+) = (\w+)\.(\w+)\)", node["Hash Cond"])
    if not m:
        raise ValueError(node["Hash Cond"])
    a1, c1, a2, c2 = m.groups()
    la = aliases(left)
    if a1 in la:
        cond = [a1, c1, a2, c2]
    else:
        cond = [a2, c2, a1, c1]
    assert cond[0] in la and cond[2] in aliases(right)
    return {"build_left": build_left, "cond": cond, "left": left, "right": right}


def main():
    d = json.load(open(os.path.join(REF, "plans.json")))
    out = {"provenance": "derived from the reference's data files plans.json, job/*.sql, job/schema.sql by scripts/extract_job_plans.py",
           "schema": schema(), "queries": {}}
    for name, plan in zip(d["names"], d["plans"]):
        sql = open(os.path.join(REF, d["sql_directory"], name + ".sql")).read()
        sel = re.findall(r"MIN\(\s*(\w+)\.(\w+)\s*\)", sql.split("FROM")[0], re.I)
        assert sel, name
        out["queries"][name] = {"select": [list(s) for s in sel], "tree": tree(plan["Plan"])}
    json.dump(out, open(OUT, "w"), separators=(",", ":"))
    print(len(out["queries"]), "queries,", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
