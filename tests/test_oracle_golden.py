"""Pin the CPU oracle against the reference's own known-answer tests
(reference tests/unit_tests.cpp, 8 cases) and its documented rules."""
import numpy as np
import pytest

import _oracle
from _golden import build_plan, expected_rows, load_cases
from pyrj import plan as pl

CASES = load_cases()


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference_unit_case(case):
    plan = build_plan(case)
    res = _oracle.execute(plan)
    exp = case["expect"]
    assert res.num_rows == exp["num_rows"]
    assert [c.type for c in res.columns] == [pl.TYPE_IDS[t] for t in exp["types"]]
    if exp["num_rows"] == 0:
        # reference returns columns with zero pages for an empty result
        assert all(c.pages.shape[0] == 0 for c in res.columns)
    assert pl.sorted_rows(res) == expected_rows(case)


def test_hash_is_murmur_fmix64():
    # src/execute.cpp:21-27 is murmur3's fmix64; known value fmix64(1)
    L = _oracle.lib()
    assert L.rjo_hash_int(0) == 0
    assert L.rjo_hash_int(1) == 0xB456BCFC34C2CB2C
    # int32 keys are sign-extended before hashing (static_cast<uint64_t>(int32))
    assert L.rjo_hash_int(-1) != L.rjo_hash_int(0xFFFFFFFF)


def test_bucket_rule():
    # src/execute.cpp:86-92 with L2 = 1 MiB (include/hardware.h:44)
    L = _oracle.lib()
    assert L.rjo_num_buckets(1, 4) == 1
    assert L.rjo_num_buckets(131072, 4) == 1  # exactly 1 MiB of (4+4)-byte entries
    assert L.rjo_num_buckets(131073, 4) == 2
    assert L.rjo_num_buckets(100_000_000, 4) == 128
    assert L.rjo_num_buckets(1_000_000_000, 4) == 128
    assert L.rjo_num_buckets(3 * 131072, 4) == 4


def test_oracle_int64_fp64_and_type_mismatch():
    # key type = build side's declared type (src/execute.cpp:271-273); probe values
    # of another variant alternative never match (:65-71)
    p = pl.Plan()
    p.new_scan_node(0, [(0, pl.INT64), (1, pl.FP64)])
    p.new_scan_node(1, [(0, pl.INT64)])
    p.new_join_node(True, 0, 1, 0, 0, [(0, pl.INT64), (1, pl.FP64), (2, pl.INT64)])
    p.new_input(pl.table_from_rows([(5, 1.5), (7, None), (None, 2.5), (2**40, -0.0)], [pl.INT64, pl.FP64]))
    p.new_input(pl.table_from_rows([(7,), (7,), (2**40,), (None,), (9,)], [pl.INT64]))
    p.root = 2
    res = _oracle.execute(p)
    assert pl.sorted_rows(res) == sorted([(7, None, 7), (7, None, 7), (2**40, -0.0, 2**40)], key=pl._sort_key)
    # INT32 probe column against INT64 build key: no rows
    q = pl.Plan()
    q.new_scan_node(0, [(0, pl.INT64)])
    q.new_scan_node(1, [(0, pl.INT32)])
    q.new_join_node(True, 0, 1, 0, 0, [(0, pl.INT64), (1, pl.INT32)])
    q.new_input(pl.table_from_rows([(1,), (2,)], [pl.INT64]))
    q.new_input(pl.table_from_rows([(1,), (2,)], [pl.INT32]))
    q.root = 2
    assert _oracle.execute(q).num_rows == 0


def test_oracle_multi_join_tree():
    # ((A ⋈ B) ⋈ C) with column reorder + duplication in output_attrs
    rng = np.random.default_rng(5)
    a = [(int(k), int(k) * 10) for k in rng.integers(0, 20, 40)]
    b = [(int(k), f"b{int(k)}") for k in rng.integers(0, 20, 30)]
    c = [(int(k),) for k in rng.integers(0, 20, 25)]
    p = pl.Plan()
    sa = p.new_scan_node(0, [(0, pl.INT32), (1, pl.INT32)])
    sb = p.new_scan_node(1, [(1, pl.VARCHAR), (0, pl.INT32)])
    sc = p.new_scan_node(2, [(0, pl.INT32)])
    j1 = p.new_join_node(False, sa, sb, 0, 1, [(1, pl.INT32), (2, pl.VARCHAR), (0, pl.INT32), (0, pl.INT32)])
    j2 = p.new_join_node(True, sc, j1, 0, 2, [(2, pl.VARCHAR), (0, pl.INT32), (1, pl.INT32), (4, pl.INT32)])
    p.new_input(pl.table_from_rows(a, [pl.INT32, pl.INT32]))
    p.new_input(pl.table_from_rows(b, [pl.INT32, pl.VARCHAR]))
    p.new_input(pl.table_from_rows(c, [pl.INT32]))
    p.root = j2
    res = _oracle.execute(p)
    exp = []
    for (ck,) in c:
        for ak, av in a:
            if ak != ck:
                continue
            for bk, bs in b:
                if bk == ak:
                    exp.append((bs.encode(), ck, av, ak))
    assert pl.sorted_rows(res) == sorted(exp, key=pl._sort_key)
