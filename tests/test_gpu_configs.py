"""BASELINE.json configs as parity cases, at sizes the oracle finishes in seconds, plus
size-independent properties at larger sizes."""
import numpy as np
import pytest

import _oracle
from pyrj import capi
from pyrj import plan as pl

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.build_context()
    yield c
    capi.destroy_context(c)


def zipf_ranks(rng, n_keys, n, s=0.9):
    """n draws of a Zipf(s) rank over [0, n_keys) by inverse-CDF on the exact weights."""
    w = 1.0 / np.arange(1, n_keys + 1, dtype=np.float64) ** s
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    return np.searchsorted(cdf, rng.random(n), side="right").astype(np.int64)


def test_config3_shape_zipf_probe_int64_payload(ctx):
    """config 3 at 2M x 6M: unique INT32 build keys + INT64 payload, Zipf-0.9 probe keys
    scattered through a fixed bijection + INT64 payload.  The hottest key owns > JN_HEAVY
    probe tuples, so the heavy-task path runs; carries are two words wide."""
    rng = np.random.default_rng(3)
    nb, npr = 2_000_000, 6_000_000
    bk = rng.permutation(nb).astype(np.int32)
    ranks = zipf_ranks(rng, nb, npr)
    pk = ((ranks * 7919 + 13) % nb).astype(np.int32)  # bijection: hot keys not adjacent
    assert np.bincount(pk.astype(np.int64), minlength=nb).max() > 70_000  # JN_HEAVY = 65536
    bt = pl.make_table([(pl.INT32, bk), (pl.INT64, rng.integers(-(2**62), 2**62, nb).astype(np.int64))])
    pt = pl.make_table([(pl.INT32, pk), (pl.INT64, rng.integers(-(2**62), 2**62, npr).astype(np.int64))])
    p = pl.Plan()
    p.new_scan_node(0, [(0, pl.INT32), (1, pl.INT64)])
    p.new_scan_node(1, [(0, pl.INT32), (1, pl.INT64)])
    p.new_join_node(True, 0, 1, 0, 0, [(0, pl.INT32), (1, pl.INT64), (3, pl.INT64)])
    p.new_input(bt)
    p.new_input(pt)
    p.root = 2
    got = capi.execute(p, ctx)
    want = _oracle.execute(p)
    assert got.num_rows == want.num_rows == npr
    assert pl.table_digest(got) == pl.table_digest(want)


def test_job_shaped_tree_with_varchar_outputs(ctx):
    """configs 1/5 in miniature: the join tree of JOB 1a (4 hash joins over company_type,
    info_type, movie_companies, movie_info_idx, title; build side = right child, as in all
    864 Hash Joins of plans.json) over synthetic IMDB-shaped tables, VARCHAR outputs.
    IMDB data and DuckDB are not available offline, so the check is GPU vs oracle."""
    rng = np.random.default_rng(5)
    n_title, n_mc, n_mi = 20_000, 60_000, 40_000

    def names(prefix, n, null_every=0):
        return [None if null_every and i % null_every == 0 else f"{prefix}-{i}".encode() for i in range(n)]

    title = pl.make_table([(pl.INT32, np.arange(n_title, dtype=np.int32)), (pl.VARCHAR, names("title", n_title)),
                           (pl.INT32, rng.integers(1900, 2020, n_title).astype(np.int32), rng.random(n_title) > 0.1)])
    ct = pl.make_table([(pl.INT32, np.arange(4, dtype=np.int32)), (pl.VARCHAR, names("kind", 4))])
    it = pl.make_table([(pl.INT32, np.arange(113, dtype=np.int32)), (pl.VARCHAR, names("info", 113))])
    mc = pl.make_table([(pl.INT32, rng.integers(0, n_title, n_mc).astype(np.int32)),
                        (pl.INT32, rng.integers(0, 6, n_mc).astype(np.int32)),
                        (pl.VARCHAR, names("note", n_mc, null_every=3))])
    mi = pl.make_table([(pl.INT32, rng.integers(0, n_title, n_mi).astype(np.int32)),
                        (pl.INT32, rng.integers(0, 113, n_mi).astype(np.int32), rng.random(n_mi) > 0.05)])
    p = pl.Plan()
    s_it = p.new_scan_node(0, [(0, pl.INT32)])
    s_mi = p.new_scan_node(1, [(0, pl.INT32), (1, pl.INT32)])
    j1 = p.new_join_node(False, s_mi, s_it, 1, 0, [(0, pl.INT32)])  # mi_idx.info_type_id = it.id
    s_mc = p.new_scan_node(2, [(0, pl.INT32), (1, pl.INT32), (2, pl.VARCHAR)])
    j2 = p.new_join_node(False, s_mc, j1, 0, 0, [(0, pl.INT32), (1, pl.INT32), (2, pl.VARCHAR)])  # mc.movie_id = mi_idx.movie_id
    s_ct = p.new_scan_node(3, [(0, pl.INT32)])
    j3 = p.new_join_node(False, j2, s_ct, 1, 0, [(0, pl.INT32), (2, pl.VARCHAR)])  # mc.company_type_id = ct.id
    s_t = p.new_scan_node(4, [(0, pl.INT32), (1, pl.VARCHAR), (2, pl.INT32)])
    j4 = p.new_join_node(False, s_t, j3, 0, 0, [(4, pl.VARCHAR), (1, pl.VARCHAR), (2, pl.INT32)])  # t.id = mc.movie_id
    for t in (it, mi, mc, ct, title):
        p.new_input(t)
    p.root = j4
    got = capi.execute(p, ctx)
    want = _oracle.execute(p)
    assert got.num_rows == want.num_rows > 10_000
    assert pl.sorted_rows(got) == pl.sorted_rows(want)


def test_full_size_properties_pk_fk(ctx):
    """20M x 50M through the resident path: |out| = |probe| (every probe key hits exactly one
    build row), key column == probe keys as a multiset (checksum of checksums), and the build
    payload is the inverse permutation of the key — no oracle involved."""
    import torch

    nb, npr = 20_000_000, 50_000_000
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    bk = torch.randperm(nb, generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    pk = torch.randint(0, nb, (npr,), generator=g, device="cuda", dtype=torch.int64).to(torch.int32)
    import bench

    R = bench.adopt(ctx, [bk, torch.arange(nb, device="cuda", dtype=torch.int32)])
    S = bench.adopt(ctx, [pk, torch.arange(npr, device="cuda", dtype=torch.int32)])
    res = ctx.execute_resident(bench.join_plan(), [R, S])
    assert res.num_rows == npr
    n_pages = res.col_pages(0)
    assert n_pages == (npr + 1983) // 1984

    def column(c):
        ptr = res.device_pages(c)
        # view the result's Page images in place (no copy) through the CUDA array interface
        class _View:
            __cuda_array_interface__ = {"shape": (n_pages, 2048), "typestr": "<i4", "data": (ptr, False), "version": 3}
        pages = torch.as_tensor(_View(), device="cuda")
        return pages[:, 1:1985].reshape(-1)[:npr]

    key, bpay, ppay = column(0), column(1), column(2)
    assert int(key.to(torch.int64).sum()) == int(pk.to(torch.int64).sum())
    assert torch.equal(torch.sort(ppay).values, torch.arange(npr, device="cuda", dtype=torch.int32))  # every probe row once
    assert torch.equal(pk[ppay.long()], key)  # the key of an output row is its probe row's key
    assert torch.equal(bk[bpay.long()], key)  # ... and its build row's key
    hdr = torch.as_tensor(type("_H", (), {"__cuda_array_interface__": {"shape": (n_pages, 2048), "typestr": "<i4", "data": (res.device_pages(0), False), "version": 3}})(), device="cuda")[:, 0]
    assert int((hdr & 0xFFFF).sum()) == npr and bool(((hdr & 0xFFFF) == (hdr >> 16)).all())
    res.free()
    R.release()
    S.release()
