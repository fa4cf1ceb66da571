"""JOB (Join Order Benchmark) plans over synthetic IMDB-shaped data.

Counterpart of the reference harness' plan loader (tests/read_sql.cpp:861-1141, rules in
SURVEY.md Appendix B) for the 113 queries of plans.json, working from the condensed fixture
tests/golden/job_plans.json (scripts/extract_job_plans.py).  The IMDB CSVs, DuckDB and the
SQL/JSON libraries the harness uses are not available offline, so:
  * join conditions come from the plan's "Hash Cond" (equivalent to the harness' DSU over
    the WHERE clause: one equi-condition per table pair);
  * build side = the child under the "Hash" node (read_sql.cpp:943-953);
  * filters are dropped (they run on the host before execute());
  * inputs are seeded synthetic tables with the IMDB schema: `id` unique, foreign keys drawn
    from the parent table's id range, nullable columns with NULLs, VARCHAR payloads.
"""
from __future__ import annotations

import json
import os

import numpy as np

from . import plan as pl

FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "job_plans.json")

# foreign-key column -> parent table (IMDB naming)
FK_PARENT = {
    "movie_id": "title", "linked_movie_id": "title", "episode_of_id": "title",
    "person_id": "name", "person_role_id": "char_name", "role_id": "role_type",
    "company_id": "company_name", "company_type_id": "company_type", "info_type_id": "info_type",
    "keyword_id": "keyword", "kind_id": "kind_type", "link_type_id": "link_type",
    "subject_id": "comp_cast_type", "status_id": "comp_cast_type",
}
# rows per table at scale 1.0 (dimension tables keep their real, tiny sizes)
BASE_ROWS = {
    "company_type": 4, "comp_cast_type": 4, "kind_type": 7, "role_type": 12, "link_type": 18, "info_type": 113,
    "title": 600, "name": 500, "char_name": 400, "company_name": 300, "keyword": 300,
    "aka_name": 500, "aka_title": 300, "cast_info": 1500, "complete_cast": 400, "movie_companies": 900,
    "movie_info": 1200, "movie_info_idx": 700, "movie_keyword": 1000, "movie_link": 300, "person_info": 800,
}


def load_fixture(path=FIXTURE):
    with open(path) as f:
        return json.load(f)


def make_tables(schema, seed=0, scale=1.0):
    """-> {table: ColumnarTable} with ALL columns of the schema (the harness loads every
    column of a scanned table, read_sql.cpp:1100-1107)."""
    rng = np.random.default_rng(seed)
    rows = {t: max(2, int(round(BASE_ROWS.get(t, 500) * (scale if BASE_ROWS.get(t, 500) > 150 else 1.0)))) for t in schema}
    out = {}
    for t, cols in schema.items():
        n = rows[t]
        spec = []
        for name, typ, nullable in cols:
            valid = rng.random(n) >= 0.07 if nullable else None
            if typ == "INT32":
                if name == "id":
                    vals = rng.permutation(n).astype(np.int32)
                elif name in FK_PARENT:
                    vals = rng.integers(0, rows[FK_PARENT[name]], n).astype(np.int32)
                else:
                    vals = rng.integers(1880, 2025, n).astype(np.int32)
                spec.append((pl.INT32, vals, valid) if valid is not None else (pl.INT32, vals))
            else:
                strs = [f"{t}.{name}.{i}".encode() for i in range(n)]
                if valid is not None:
                    strs = [s if v else None for s, v in zip(strs, valid)]
                spec.append((pl.VARCHAR, strs))
        out[t] = pl.make_table(spec)
    return out


REAL_ROWS = {  # IMDB (JOB snapshot) cardinalities
    "aka_name": 901343, "aka_title": 361472, "cast_info": 36244344, "char_name": 3140339, "comp_cast_type": 4,
    "company_name": 234997, "company_type": 4, "complete_cast": 135086, "info_type": 113, "keyword": 134170,
    "kind_type": 7, "link_type": 18, "movie_companies": 2609129, "movie_info": 14835720, "movie_info_idx": 1380035,
    "movie_keyword": 4523930, "movie_link": 29997, "name": 4167491, "person_info": 2963664, "role_type": 12,
    "title": 2528312,
}


def make_scan_table(schema_cols, table, n, rng):
    """One scan's (already filtered) input at realistic size, generated with vectorised numpy
    only: `id` a random subset of the real id range, foreign keys uniform over the parent's real
    id range, nullable INT32 columns with 5 % NULLs, VARCHAR columns fixed-length strings."""
    from . import pages as pg

    t = pl.ColumnarTable(n, [])
    for name, typ, nullable in schema_cols:
        if typ == "INT32":
            if name == "id":
                vals = rng.choice(REAL_ROWS[table], size=n, replace=False).astype(np.int32) if n < REAL_ROWS[table] else rng.permutation(n).astype(np.int32)
            elif name in FK_PARENT:
                vals = rng.integers(0, REAL_ROWS[FK_PARENT[name]], n).astype(np.int32)
            else:
                vals = rng.integers(1880, 2025, n).astype(np.int32)
            valid = (rng.random(n) >= 0.05) if nullable else None
            t.columns.append(pl.Column(pl.INT32, pg.pack_fixed(vals, valid, pl.INT32)))
        else:
            t.columns.append(pl.Column(pl.VARCHAR, pg.pack_varchar_fixed(np.arange(n), digits=9, prefix=name[:3].encode())))
    return t


def make_scaled_inputs(query, schema, rng, cache):
    """alias -> ColumnarTable for every scan of `query`, sized by PostgreSQL's "Plan Rows"
    estimate of that scan (x3 parallel workers when filtered, capped by the real cardinality;
    the real cardinality when unfiltered).  `cache` shares tables between queries."""
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


def _aliases(tree):
    return {tree["alias"]} if "scan" in tree else _aliases(tree["left"]) | _aliases(tree["right"])


def build_plan(query, schema, tables, by_alias: bool = False) -> pl.Plan:
    """The Plan the harness would hand to Contest::execute for this query.  `tables` maps base
    table name -> ColumnarTable (or alias -> ColumnarTable with by_alias=True: one separately
    filtered input per scan, as the harness produces)."""
    plan = pl.Plan()
    col_index = {t: {c[0]: i for i, c in enumerate(cols)} for t, cols in schema.items()}
    col_type = {t: {c[0]: pl.TYPE_IDS[c[1]] for c in cols} for t, cols in schema.items()}
    alias_table = {}

    def collect(tree):
        if "scan" in tree:
            alias_table[tree["alias"]] = tree["scan"]
        else:
            collect(tree["left"])
            collect(tree["right"])

    collect(query["tree"])

    def typ(attr):
        return col_type[alias_table[attr[0]]][attr[1]]

    def walk(tree, required):
        """-> (node index, output attr list); nodes are appended post-order, left first
        (read_sql.cpp:1008-1009,1056-1062)."""
        if "scan" in tree:
            table = tree["scan"]
            inp = plan.new_input(tables[tree["alias"]] if by_alias else tables[table])
            outs = [(col_index[table][c], typ((a, c))) for a, c in required]
            return plan.new_scan_node(inp, outs), list(required)
        la = _aliases(tree["left"])
        lkey, rkey = (tree["cond"][0], tree["cond"][1]), (tree["cond"][2], tree["cond"][3])
        lreq = [r for r in required if r[0] in la]
        rreq = [r for r in required if r[0] not in la]
        # a side's own key column is appended if not already required (read_sql.cpp:981-1007)
        if lkey not in lreq:
            lreq.append(lkey)
        if rkey not in rreq:
            rreq.append(rkey)
        ln, lout = walk(tree["left"], lreq)
        rn, rout = walk(tree["right"], rreq)
        both = lout + rout
        outs = [(both.index(r), typ(r)) for r in required]
        node = plan.new_join_node(tree["build_left"], ln, rn, lout.index(lkey), rout.index(rkey), outs)
        return node, list(required)

    required = [tuple(s) for s in query["select"]]
    root, _ = walk(query["tree"], required)
    plan.root = root
    return plan
