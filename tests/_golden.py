"""Load tests/golden/unit_cases.json into pyrj plans."""
import json
import os

from pyrj import plan as pl

HERE = os.path.dirname(os.path.abspath(__file__))


def load_cases():
    with open(os.path.join(HERE, "golden", "unit_cases.json")) as f:
        return json.load(f)["cases"]


def _attrs(out):
    return [(i, pl.TYPE_IDS[t]) for i, t in out]


def build_plan(case) -> pl.Plan:
    p = pl.Plan()
    for n in case["nodes"]:
        if "scan" in n:
            p.new_scan_node(n["scan"]["base_table_id"], _attrs(n["out"]))
        else:
            j = n["join"]
            p.new_join_node(j["build_left"], j["left"], j["right"], j["left_attr"], j["right_attr"], _attrs(n["out"]))
    for inp in case["inputs"]:
        types = [pl.TYPE_IDS[t] for t in inp["types"]]
        rows = [tuple(r) for r in inp["rows"]]
        if rows:
            p.new_input(pl.table_from_rows(rows, types))
        else:
            # reference "Empty join": typed columns with zero pages, num_rows 0
            p.new_input(pl.ColumnarTable(0, [pl.Column(t) for t in types]))
    p.root = case["root"]
    return p


def expected_rows(case):
    rows = []
    for r in case["expect"]["rows"]:
        rows.append(tuple(v.encode() if isinstance(v, str) else v for v in r))
    return sorted(rows, key=pl._sort_key)
