"""Host page codec (pyrj.pages) vs the oracle's restatement of the reference's
encoder (Table::to_columnar, build_table.cpp:456-681) and decoder
(Table::from_columnar, build_table.cpp:312-436)."""
import numpy as np
import pytest

import _oracle
from pyrj import pages as pg
from pyrj import plan as pl


def test_full_page_capacities():
    # 1984 INT32 / 1007 INT64 values per page without NULLs
    assert pg.rows_per_full_page(pg.INT32) == 1984
    assert pg.rows_per_full_page(pg.INT64) == 1007
    assert pg.rows_per_full_page(pg.FP64) == 1007


@pytest.mark.parametrize("dtype", [pg.INT32, pg.INT64, pg.FP64])
@pytest.mark.parametrize("n,null_frac", [(0, 0.0), (1, 0.0), (1984, 0.0), (1985, 0.0), (5000, 0.0), (5000, 0.3), (30000, 0.97), (70000, 1.0), (3, 1.0)])
def test_pack_matches_reference_fill_rule(dtype, n, null_frac):
    rng = np.random.default_rng(n * 7 + dtype)
    if dtype == pg.FP64:
        vals = rng.standard_normal(n)
    else:
        info = np.iinfo(pg.NP_DTYPE[dtype])
        vals = rng.integers(info.min, info.max, n, dtype=pg.NP_DTYPE[dtype], endpoint=True)
    valid = rng.random(n) >= null_frac
    ours = pg.pack_fixed(vals, valid, dtype)
    ref = _oracle.encode_fixed(dtype, vals, valid).pages
    assert ours.shape == ref.shape
    # headers and bitmaps must agree page by page; value area up to n_nonnull
    for p in range(ours.shape[0]):
        nr, nv = ours[p, :4].view(np.uint16)
        assert (nr, nv) == tuple(ref[p, :4].view(np.uint16))
        hdr, w = pg.HDR[dtype], pg.WIDTH[dtype]
        assert np.array_equal(ours[p, hdr : hdr + nv * w], ref[p, hdr : hdr + nv * w])
        nb = (int(nr) + 7) // 8
        assert np.array_equal(ours[p, pg.PAGE_SIZE - nb :], ref[p, pg.PAGE_SIZE - nb :])
    # both decoders agree on both encodings
    v1, m1 = pg.unpack_fixed(ours, n, dtype)
    v2, m2 = _oracle.decode_fixed(pl.Column(dtype, ours), n)
    assert np.array_equal(m1, valid) and np.array_equal(m2, valid)
    assert np.array_equal(v1[valid], vals[valid]) and np.array_equal(v2[valid], vals[valid])


def test_decode_row_overflow_raises():
    pages = pg.pack_fixed(np.arange(10, dtype=np.int32), None, pg.INT32)
    with pytest.raises(RuntimeError, match="row_idx"):
        pg.unpack_fixed(pages, 5, pg.INT32)
    with pytest.raises(RuntimeError, match="row_idx"):
        _oracle.decode_fixed(pl.Column(pg.INT32, pages), 5)


def test_varchar_roundtrip_and_long_strings():
    rng = np.random.default_rng(3)
    strings = []
    for i in range(600):
        r = rng.random()
        if r < 0.15:
            strings.append(None)
        elif r < 0.17:
            strings.append(bytes(rng.integers(97, 123, int(rng.integers(8186, 30000)), dtype=np.uint8)))
        else:
            strings.append(bytes(rng.integers(32, 127, int(rng.integers(0, 200)), dtype=np.uint8)))
    strings += [b"", None, b"x" * 8185, b"y" * 8186, b"z" * 8188, b"w" * 8189]
    ours = pg.pack_varchar(strings)
    ref = _oracle.encode_varchar(strings).pages
    assert ours.shape == ref.shape
    assert np.array_equal(ours[:, :4], ref[:, :4])
    n = len(strings)
    assert pg.unpack_varchar(ours, n) == strings
    assert _oracle.decode_varchar(pl.Column(pg.VARCHAR, ours), n) == strings
    assert pg.unpack_varchar(ref, n) == strings


def test_row_idx_rule_follows_the_reference_loop():
    """reference src/build_table.cpp:332-343: "row_idx" only when a NON-NULL value lands at a
    row index >= num_rows; NULL rows past the end are tolerated, and fixed-width pages are
    decoded from the bitmap alone (the header's non-null count is never read)"""
    vals = np.arange(10, dtype=np.int32)
    valid = np.ones(10, bool)
    valid[7:] = False  # rows 7..9 NULL
    pages = pg.pack_fixed(vals, valid, pg.INT32)
    for decode in (lambda p, n: pg.unpack_fixed(p, n, pg.INT32), lambda p, n: _oracle.decode_fixed(pl.Column(pg.INT32, p), n)):
        v, m = decode(pages, 7)  # three trailing NULL rows beyond num_rows: fine
        assert m.all() and np.array_equal(v, vals[:7])
        with pytest.raises(RuntimeError, match="row_idx"):
            decode(pages, 6)  # row 6 is non-NULL and lands past the end
    # a header that lies about the non-null count changes nothing: the bitmap decides
    lying = pages.copy()
    lying[0, 2:4].view(np.uint16)[0] = 10
    for decode in (lambda p, n: pg.unpack_fixed(p, n, pg.INT32), lambda p, n: _oracle.decode_fixed(pl.Column(pg.INT32, p), n)):
        v, m = decode(lying, 10)
        assert np.array_equal(m, valid) and np.array_equal(v[valid], vals[valid])
