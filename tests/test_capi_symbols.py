"""The C-ABI library loads and exports every symbol include/rj.h declares
(no compute calls: there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "radix-join_amd", "librj.so")


def _declared():
    src = open(os.path.join(ROOT, "include", "rj.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(rj_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__ as g

        g.build()
    return ctypes.CDLL(LIB)


def test_header_declares_the_documented_entry_points():
    from pyrj import capi

    assert sorted(capi.EXPORTS) == _declared()


def test_library_exports_every_declared_symbol(lib):
    for name in _declared():
        assert hasattr(lib, name), f"librj.so does not export {name}"
    lib.rj_abi_version.restype = ctypes.c_int
    assert lib.rj_abi_version() == 3


def test_no_gpu_fails_loudly_without_fallback(lib):
    """On a box without a GPU the product path must refuse to run (no CPU fallback)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pyrj import capi

    with pytest.raises(capi.RjError) as e:
        capi.Context()
    assert e.value.code == 6  # RJ_ERR_NO_GPU
    assert "no CPU fallback" in e.value.message


def test_product_never_touches_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may use oracle/."""
    pkg = os.path.join(ROOT, "radix-join_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "rjo_" not in txt and "librjo" not in txt and "_oracle" not in txt, f


def test_plan_shardable_is_decided_from_the_plan_alone():
    """rj_plan_shardable needs neither a context nor a GPU: per side of a JoinNode, what must travel
    with the key has to fit the carry words (3 words behind an INT32 key, 2 behind a 64-bit one; at
    most one 64-bit column), and no scan may output VARCHAR"""
    from pyrj import capi
    from pyrj import plan as pl

    def plan(b_out, p_out, j_out):
        p = pl.Plan()
        p.new_scan_node(0, b_out)
        p.new_scan_node(1, p_out)
        p.new_join_node(True, 0, 1, 0, 0, j_out)
        p.root = 2
        return p

    i32, i64, vc = pl.INT32, pl.INT64, pl.VARCHAR
    ok, why = capi.plan_shardable(plan([(0, i32), (1, i64)], [(0, i32), (1, i32)], [(0, i32), (1, i64), (3, i32)]))
    assert ok and why == ""
    ok, why = capi.plan_shardable(plan([(0, i32), (1, i64), (2, i32)], [(0, i32)], [(1, i64), (2, i32)]))
    assert ok and why == ""  # INT64 + INT32 = three carry words
    ok, why = capi.plan_shardable(plan([(0, i32), (1, i32), (2, i32), (3, i32)], [(0, i32)], [(1, i32), (2, i32), (3, i32)]))
    assert ok
    ok, why = capi.plan_shardable(plan([(0, i32), (1, i64), (2, i64)], [(0, i32)], [(1, i64), (2, i64)]))
    assert not ok and "more payload than travels with the key" in why
    ok, why = capi.plan_shardable(plan([(0, i64), (1, i64), (2, i32)], [(0, i64)], [(1, i64), (2, i32)]))
    assert not ok  # a 64-bit key leaves two carry words
    ok, why = capi.plan_shardable(plan([(0, i32), (1, vc)], [(0, i32)], [(1, vc)]))
    assert not ok and "VARCHAR" in why
    # two joins, each within the limit
    p = pl.Plan()
    a = p.new_scan_node(0, [(0, i32), (1, i32)])
    b = p.new_scan_node(1, [(0, i32)])
    j1 = p.new_join_node(True, a, b, 0, 0, [(0, i32), (1, i32)])
    c = p.new_scan_node(2, [(0, i32), (1, i64)])
    p.root = p.new_join_node(False, j1, c, 1, 0, [(0, i32), (3, i64), (1, i32)])
    assert capi.plan_shardable(p)[0]


def test_every_environment_switch_is_documented():
    """every RJ_* variable the library or the shim reads appears in INTEGRATION.md's table"""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for sub in ("radix-join_amd/csrc", "radix-join_amd/host"):
        d = os.path.join(root, sub)
        for fn in os.listdir(d):
            if fn.endswith((".hip", ".cpp", ".hpp")):
                names |= set(re.findall(r'(?:env_int|getenv)\("(RJ_[A-Z0-9_]+)"', open(os.path.join(d, fn)).read()))
    assert len(names) > 20
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing


def test_bench_and_smoke_refuse_to_run_without_a_gpu():
    """bench.py and __graft_entry__.smoke() measure / check the HIP path only: on a box without a GPU
    they stop with a message instead of timing or checking anything else (skipped where a GPU is present)."""
    import os
    import subprocess
    import sys

    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("a GPU is present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-extras", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout), r.stderr[-500:]
    assert not any(line.startswith("{") for line in r.stdout.splitlines()), "no result line without a GPU"
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode != 0, "smoke() must fail without a GPU"
