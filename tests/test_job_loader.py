"""Plan loader for the JOB workload (pyrj.job) — CPU tier: fixture shape, the harness'
attribute-propagation rules and the oracle on every plan."""
import pytest

import _oracle
from pyrj import job
from pyrj import plan as pl

FX = job.load_fixture()
TABLES = job.make_tables(FX["schema"], seed=1)


def test_fixture_matches_the_corpus_facts():
    # 113 queries, 864 inner hash joins, 21 tables,
    # 59 INT32 + 49 VARCHAR columns, every build side is the right child
    assert len(FX["queries"]) == 113
    assert len(FX["schema"]) == 21
    cols = [c for t in FX["schema"].values() for c in t]
    assert sum(1 for c in cols if c[1] == "INT32") == 59 and sum(1 for c in cols if c[1] == "VARCHAR") == 49

    def joins(t):
        return [] if "scan" in t else [t] + joins(t["left"]) + joins(t["right"])

    all_joins = [j for q in FX["queries"].values() for j in joins(q["tree"])]
    assert len(all_joins) == 864
    assert not any(j["build_left"] for j in all_joins)
    assert max(len(joins(q["tree"])) for q in FX["queries"].values()) == 16


def test_plan_1a_follows_the_harness_rules():
    p = job.build_plan(FX["queries"]["1a"], FX["schema"], TABLES)
    # 5 scans + 4 joins, appended post-order, root last (read_sql.cpp:1056-1062,1139)
    assert len(p.nodes) == 9 and p.root == 8 and len(p.inputs) == 5
    root = p.nodes[p.root]
    # SELECT MIN(mc.note), MIN(t.title), MIN(t.production_year)  (job/1a.sql)
    assert [a[1] for a in root.output_attrs] == [pl.VARCHAR, pl.VARCHAR, pl.INT32]
    assert isinstance(root.data, pl.JoinNode) and root.data.build_left is False
    # left child of the root is the scan of title: required {title, production_year} + key id
    left = p.nodes[root.data.left]
    assert isinstance(left.data, pl.ScanNode)
    t_cols = [c[0] for c in FX["schema"]["title"]]
    assert [t_cols[i] for i, _ in left.output_attrs] == ["title", "production_year", "id"]
    assert root.data.left_attr == 2  # t.id is the appended key column
    # every input carries ALL columns of its base table (read_sql.cpp:1100-1107)
    assert len(p.inputs[0].columns) == len(FX["schema"]["title"])


@pytest.mark.parametrize("name", sorted(FX["queries"]))
def test_oracle_runs_every_plan(name):
    p = job.build_plan(FX["queries"][name], FX["schema"], TABLES)
    r = _oracle.execute(p)
    root = p.nodes[p.root]
    assert [c.type for c in r.columns] == [a[1] for a in root.output_attrs]
    assert r.num_rows > 0
