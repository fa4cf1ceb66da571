"""Synthetic BASELINE workloads built on the device, and closed-form verification of their
join results (no oracle involved: the relations are PK-FK by construction, so every property
of the result can be recomputed from the inputs).

Shared by bench.py (which verifies the result of its timed loop) and the full-size GPU tests.
Workload shapes follow SURVEY.md §8(d):

    config2     100 M ⋈ 100 M, INT32 keys (build = permutation, probe = uniform iid), one INT32
                payload column per side
    uniform1b   the same at 1 B ⋈ 1 B
    config3     1 B ⋈ 1 B, build = permutation, probe = Zipf-0.9 ranks scattered through a fixed
                bijection (skew on the probe side only), one INT64 payload column per side

Payload of row i is i (INT32) or i * PAY_MUL (INT64): the build/probe row an output row came
from can be read back out of the result.
"""
from __future__ import annotations

import torch

from . import plan as pl

ROWS32 = 1984  # rows of a full non-NULL INT32 page (reference src/build_table.cpp:488)
ROWS64 = 1007  # rows of a full non-NULL INT64/FP64 page (reference src/build_table.cpp:531)
PAY_MUL = 1_000_003

WORKLOADS = {
    "config2": dict(rows=100_000_000, zipf=0.0, payload64=False,
                    label="BASELINE config 2: single JoinNode, 100M x 100M INT32 uniform keys, 1 INT32 payload col per side"),
    "uniform1b": dict(rows=1_000_000_000, zipf=0.0, payload64=False,
                      label="single JoinNode, 1B x 1B INT32 uniform keys, 1 INT32 payload col per side"),
    "config3": dict(rows=1_000_000_000, zipf=0.9, payload64=True,
                    label="BASELINE config 3: single JoinNode, 1B x 1B INT32 keys, Zipf-0.9 probe keys, 1 INT64 payload col per side (skew + 2-pass radix)"),
}


# ------------------------------------------------------------------ page images on the device
def pack_pages_gpu(values: torch.Tensor) -> torch.Tensor:
    """INT32 values (device) -> Page images uint8[n_pages, 8192] on the device, no NULLs
    (layout: reference src/build_table.cpp:472-481; 1984 rows per full page)."""
    n = values.numel()
    npages = (n + ROWS32 - 1) // ROWS32
    pages = torch.zeros((npages, 2048), dtype=torch.int32, device=values.device)
    full = n // ROWS32
    if full:
        pages[:full, 1 : 1 + ROWS32] = values[: full * ROWS32].view(full, ROWS32)
        pages[:full, 0] = ROWS32 | (ROWS32 << 16)
        pages[:full, 1986:] = -1  # 248 bitmap bytes, all rows valid
    rem = n - full * ROWS32
    if rem:
        pages[full, 1 : 1 + rem] = values[full * ROWS32 :]
        pages[full, 0] = rem | (rem << 16)
        b = pages[full].view(torch.uint8)
        nb = (rem + 7) // 8
        bm = torch.full((nb,), 255, dtype=torch.uint8, device=values.device)
        if rem % 8:
            bm[-1] = (1 << (rem % 8)) - 1
        b[8192 - nb :] = bm
    return pages.view(torch.uint8).view(npages, 8192)


def pack_pages_gpu64(values: torch.Tensor) -> torch.Tensor:
    """INT64 values (device) -> Page images, 1007 rows per full page, values from byte 8
    (layout: reference src/build_table.cpp:515-524)."""
    n = values.numel()
    npages = (n + ROWS64 - 1) // ROWS64
    pages = torch.zeros((npages, 1024), dtype=torch.int64, device=values.device)
    full = n // ROWS64
    if full:
        pages[:full, 1 : 1 + ROWS64] = values[: full * ROWS64].view(full, ROWS64)
    rem = n - full * ROWS64
    if rem:
        pages[full, 1 : 1 + rem] = values[full * ROWS64 :]
    b = pages.view(torch.uint8).view(npages, 8192)
    cnt = torch.full((npages,), ROWS64, dtype=torch.int32, device=values.device)
    if rem:
        cnt[-1] = rem
    hdr = (cnt | (cnt << 16)).view(torch.uint8).view(npages, 4)
    b[:, :4] = hdr
    nbf = (ROWS64 + 7) // 8  # 126 bitmap bytes of a full page, last byte has 7 valid bits
    if full:
        b[:full, 8192 - nbf :] = 255
        b[:full, 8191] = (1 << (ROWS64 % 8)) - 1
    if rem:
        nb = (rem + 7) // 8
        b[full, 8192 - nb :] = 255
        if rem % 8:
            b[full, 8191] = (1 << (rem % 8)) - 1
    return b


def adopt(ctx, cols):
    """Device tensors -> Page images -> rj_table (resident, zero copy)."""
    pages = [pack_pages_gpu64(c) if c.dtype == torch.int64 else pack_pages_gpu(c) for c in cols]
    types = [pl.INT64 if c.dtype == torch.int64 else pl.INT32 for c in cols]
    torch.cuda.synchronize()
    n = cols[0].numel()
    return ctx.adopt_device(n, types, [p.data_ptr() for p in pages], [p.shape[0] for p in pages], keep=pages)


def join_plan(payload=None):
    """Scan(R){0,1}, Scan(S){0,1}, Join(build_left=true, left_attr=0, right_attr=0, out={0,1,3})
    (SURVEY.md §8d)."""
    payload = pl.INT32 if payload is None else payload
    p = pl.Plan()
    p.new_scan_node(0, [(0, pl.INT32), (1, payload)])
    p.new_scan_node(1, [(0, pl.INT32), (1, payload)])
    p.new_join_node(True, 0, 1, 0, 0, [(0, pl.INT32), (1, payload), (3, payload)])
    p.root = 2
    return p


# ------------------------------------------------------------------------------- relations
def zipf_keys(n_keys, n, s, device, gen):
    """n draws of a Zipf(s) rank over [0, n_keys), scattered through a fixed bijection so that hot
    keys are not numerically adjacent (skew on the probe side only)."""
    w = torch.arange(1, n_keys + 1, device=device, dtype=torch.float64).pow_(-s)
    cdf = torch.cumsum(w, 0)
    del w
    cdf /= cdf[-1].clone()
    out = torch.empty(n, device=device, dtype=torch.int64)
    step = 1 << 27
    for i in range(0, n, step):
        m = min(step, n - i)
        u = torch.rand(m, device=device, dtype=torch.float64, generator=gen)
        out[i : i + m] = torch.searchsorted(cdf, u, right=True).clamp_(max=n_keys - 1)
    del cdf
    return ((out * 7919 + 13) % n_keys).to(torch.int32)


BIJ_A, BIJ_B = 2654435761, 12345  # key of global build row i when the build side is sharded


def build_key_of_row(idx: torch.Tensor, total: int) -> torch.Tensor:
    """Multiplicative bijection of [0, total) (A is prime, so coprime to total = 2^a 5^b)."""
    return (idx * BIJ_A + BIJ_B) % total


class Relations:
    """One rank's share of a workload: rk/rp = build keys/payloads, sk/sp = probe keys/payloads
    (device tensors), n = rows per relation on this rank, total = rows per relation overall."""

    def __init__(self, name, n, total, rank, world, rk, rp, sk, sp, payload64):
        self.name, self.n, self.total, self.rank, self.world = name, n, total, rank, world
        self.rk, self.rp, self.sk, self.sp, self.payload64 = rk, rp, sk, sp, payload64

    @property
    def payload_type(self):
        return pl.INT64 if self.payload64 else pl.INT32


def make_relations(name, device, rows=None, rank=0, world=1, sharded=None) -> Relations:
    """rows = rows per relation OVERALL (default: the workload's own size); a rank of a
    `world`-way run holds the contiguous row range [rank, rank+1) * rows / world of both.
    sharded (default: world > 1): build keys are the closed-form bijection of the global row
    index (checkable on any rank) instead of a random permutation."""
    w = WORKLOADS[name]
    total = int(rows or w["rows"])
    lo, hi = total * rank // world, total * (rank + 1) // world
    n = hi - lo
    g = torch.Generator(device=device)
    if sharded is None:
        sharded = world > 1
    if not sharded:
        g.manual_seed(1)
        rk = torch.randperm(total, generator=g, device=device, dtype=torch.int64).to(torch.int32)
    else:
        idx = torch.arange(lo, hi, device=device, dtype=torch.int64)
        rk = build_key_of_row(idx, total).to(torch.int32)
        del idx
    if w["zipf"]:
        g.manual_seed(3 + rank)
        sk = zipf_keys(total, n, w["zipf"], device, g)
    else:
        g.manual_seed(2 + rank)
        sk = torch.randint(0, total, (n,), generator=g, device=device, dtype=torch.int64).to(torch.int32)
    row = torch.arange(lo, hi, device=device, dtype=torch.int64)
    if w["payload64"]:
        rp = row * PAY_MUL
        sp = rp.clone()
    else:
        rp = row.to(torch.int32)
        sp = rp.clone()
    del row
    return Relations(name, n, total, rank, world, rk, rp, sk, sp, bool(w["payload64"]))


# ----------------------------------------------------------------- views of a result in HBM
def _view(ptr, shape, typestr):
    class _V:
        __cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 3}

    return torch.as_tensor(_V(), device="cuda")


def result_column(res, c, n_rows):
    """Column c of a resident result as a strided tensor view of its Page images (no copy) plus
    the page-header words."""
    n_pages = res.col_pages(c)
    ptr = res.device_pages(c)
    if res.col_type(c) == pl.INT32:
        assert n_pages == (n_rows + ROWS32 - 1) // ROWS32, (n_pages, n_rows)
        pages = _view(ptr, (n_pages, 2048), "<i4")
        return pages[:, 1 : 1 + ROWS32], pages[:, 0]
    assert n_pages == (n_rows + ROWS64 - 1) // ROWS64, (n_pages, n_rows)
    pages = _view(ptr, (n_pages, 1024), "<i8")
    return pages[:, 1 : 1 + ROWS64], _view(ptr, (n_pages, 2048), "<i4")[:, 0]


def _mix64(x: torch.Tensor) -> torch.Tensor:
    """A cheap 64-bit mixer (wrapping int64 arithmetic) for order-independent checksums."""
    x = x * -7046029254386353131  # 0x9E3779B97F4A7C15 as int64
    x = x ^ (x >> 29)
    return x * -4658895280553007687  # 0xBF58476D1CE4E5B9


def verify_pk_fk(res, rel: Relations, chunk_pages: int = 1 << 16):
    """Closed-form check of Join(R, S) for a PK-FK workload whose R and S both live on this
    device (world == 1): every probe row appears exactly once, with its own key and payload,
    next to the one build row that holds that key.  Raises AssertionError; returns a digest
    {rows, sum_key, sum_mix_probe_payload} that is also computable from the inputs alone."""
    assert rel.world == 1, "verify_pk_fk needs both whole relations on this device"
    n = rel.n
    assert res.num_rows == n, f"result has {res.num_rows} rows, expected |S| = {n}"
    assert res.num_cols == 3
    key2d, khdr = result_column(res, 0, n)
    bp2d, bhdr = result_column(res, 1, n)
    pp2d, phdr = result_column(res, 2, n)
    for hdr, rf in ((khdr, ROWS32), (bhdr, ROWS64 if rel.payload64 else ROWS32), (phdr, ROWS64 if rel.payload64 else ROWS32)):
        nr = hdr & 0xFFFF
        assert int(nr.sum()) == n, "page headers do not add up to the row count"
        assert bool((nr == ((hdr >> 16) & 0xFFFF)).all()), "non-null count differs from row count"
        assert bool((nr[:-1] == rf).all()), "result pages are not full"
    mul = PAY_MUL if rel.payload64 else 1
    seen = torch.zeros(n, dtype=torch.uint8, device=key2d.device)
    sum_key = 0
    sum_mix = 0
    # The three columns page differently (1984 vs 1007 rows per page): walk the result in row
    # chunks and cut each column's rows out of its own pages.
    rows_per_chunk = chunk_pages * ROWS32
    for r0 in range(0, n, rows_per_chunk):
        r1 = min(n, r0 + rows_per_chunk)

        def rows_of(col2d, rf):
            p0, p1 = r0 // rf, (r1 + rf - 1) // rf
            flat = col2d[p0:p1].reshape(-1)
            return flat[r0 - p0 * rf : r1 - p0 * rf]

        key = rows_of(key2d, ROWS32).to(torch.int64)
        bpay = rows_of(bp2d, ROWS64 if rel.payload64 else ROWS32).to(torch.int64)
        ppay = rows_of(pp2d, ROWS64 if rel.payload64 else ROWS32).to(torch.int64)
        if mul != 1:
            assert bool((bpay % mul == 0).all()) and bool((ppay % mul == 0).all()), "payload is not a row multiple"
        brow, prow = bpay // mul, ppay // mul
        assert bool(((prow >= 0) & (prow < n)).all()) and bool(((brow >= 0) & (brow < n)).all()), "row id out of range"
        assert bool((rel.sk[prow].to(torch.int64) == key).all()), "output key differs from its probe row's key"
        assert bool((rel.rk[brow].to(torch.int64) == key).all()), "output key differs from its build row's key"
        seen[prow] = 1
        sum_key += int(key.sum())
        sum_mix = (sum_mix + int(_mix64(ppay).sum())) & 0xFFFFFFFFFFFFFFFF
        del key, bpay, ppay, brow, prow
    assert bool(seen.all()), "some probe row is missing from the result (or another one is duplicated)"
    del seen
    want_key = int(rel.sk.to(torch.int64).sum())
    assert sum_key == want_key, "sum of output keys differs from the sum of probe keys"
    want_mix = 0
    step = 1 << 27
    for i in range(0, n, step):
        want_mix = (want_mix + int(_mix64(rel.sp[i : i + step].to(torch.int64)).sum())) & 0xFFFFFFFFFFFFFFFF
    assert sum_mix == want_mix, "checksum of output probe payloads differs from the probe side's"
    return {"rows": n, "sum_key": sum_key, "sum_mix_probe_payload": sum_mix}
