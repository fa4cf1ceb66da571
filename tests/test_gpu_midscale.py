"""Mid-scale differential test: random single- and two-join plans over 0.3-4 M-row tables —
large enough for the multi-pass partitioner with realistically filled partitions (two passes,
fine histogram, LDS tables at their working load), small enough for the oracle to finish in a
couple of seconds each.  Duplicate build keys, NULL keys, hot probe keys, INT32/INT64/FP64 keys
and payloads, nullable payloads.  Compared by order-independent digest of the result rows."""
import numpy as np
import pytest

import _oracle
from pyrj import capi
from pyrj import plan as pl

pytestmark = pytest.mark.gpu

NP_OF = {pl.INT32: np.int32, pl.INT64: np.int64, pl.FP64: np.float64}


# "xcd": the XCD-aware output placement of the big passes (per-XCD sub-ranges of every partition
# in pass 1, grid transposition in the later passes; csrc/rj_device.hpp PassParams::xcd_log2 /
# xcd_remap) normally starts at 40 Mi tuples — here it is forced for every pass, so these
# 0.3-4 M-row joins (several tile groups per segment) run through it against the oracle.
# "side": on top of that, 16 forced radix bits (two plain-histogram passes) with the digit side
# arrays between the passes switched on (RJ_TUNE_AOS_MID=1 for 12-byte tuples, RJ_TUNE_PACKED_SIDE=1
# for packed pairs; both measured net-neutral and off by default): the later pass' histogram then
# reads 16-bit digits with the vector loader, full tiles at odd offsets included.
_KNOBS = {
    "default": ({}, {}),
    "xcd": ({"RJ_TUNE_XCD_MIN_ROWS": "0"}, {}),
    "side": ({"RJ_TUNE_XCD_MIN_ROWS": "0", "RJ_TUNE_AOS_MID": "1", "RJ_TUNE_PACKED_SIDE": "1"}, {"radix_bits": 16}),
    # 12-byte tuples between the passes WITHOUT the side array: the later histogram reads the keys
    # out of the tuples (Aos3KeyLoader)
    "mid3": ({"RJ_TUNE_XCD_MIN_ROWS": "0", "RJ_TUNE_AOS_MID": "3"}, {"radix_bits": 16}),
    # the second pass of packed two-pass plans chunk by chunk on two streams (RJ_TUNE_MALL_CHUNK: measured
    # slower at 1 B rows and off by default; 3 segments per chunk here, so that chunks end inside the XCD grid)
    "chunks": ({"RJ_TUNE_XCD_MIN_ROWS": "0", "RJ_TUNE_MALL_CHUNK": "3"}, {"radix_bits": 16}),
    # packed pairs between the passes as plain 8-byte pairs (RJ_TUNE_BLOCKED_MID=0; the default since round 3
    # is blocks of 256 keys + 256 carries, which the 16-bit knobs above run through for key + INT32-payload plans)
    "pairs": ({"RJ_TUNE_XCD_MIN_ROWS": "0", "RJ_TUNE_BLOCKED_MID": "0"}, {"radix_bits": 16}),
}


@pytest.fixture(scope="module", params=list(_KNOBS))
def ctx(request):
    import os

    env, kw = _KNOBS[request.param]
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)  # read once, when the context is created
    try:
        c = capi.Context(**kw)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    yield c
    capi.destroy_context(c)


def keys(rng, n, domain, dt, hot=0.0):
    k = rng.integers(0, domain, n)
    if hot:  # one key owns a fraction `hot` of the rows (heavy-task path on the probe side)
        k[rng.random(n) < hot] = domain // 3
    if dt == pl.FP64:
        return k.astype(np.float64) * 0.5 - 1000.0
    if dt == pl.INT64:
        return k.astype(np.int64) * 4_000_000_007 - 12345  # needs the high word
    return k.astype(np.int32) - 7


def column(rng, n, dt, null_frac):
    if dt == pl.FP64:
        v = rng.standard_normal(n)
    elif dt == pl.INT64:
        v = rng.integers(-(2**62), 2**62, n)
    else:
        v = rng.integers(-(2**31), 2**31 - 1, n)
    v = v.astype(NP_OF[dt])
    if null_frac:
        return (dt, v, rng.random(n) >= null_frac)
    return (dt, v)


@pytest.mark.parametrize("seed", range(20))
def test_midscale_random_join(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    kt = [pl.INT32, pl.INT32, pl.INT64, pl.FP64][seed % 4]
    nb = int(rng.integers(300_000, 3_000_000))
    npr = int(rng.integers(500_000, 4_000_000))
    # average build multiplicity between 1 (domain 2x) and 2 (domain 0.5x): output <= ~2 |probe|
    domain = int(nb * rng.uniform(0.5, 2.0))
    bk = keys(rng, nb, domain, kt)
    pk = keys(rng, npr, domain, kt, hot=0.1 if seed % 3 == 0 else 0.0)
    null_b = 0.05 if seed % 2 else 0.0
    null_p = 0.08 if seed % 4 == 1 else 0.0
    bkey = (kt, bk, rng.random(nb) >= null_b) if null_b else (kt, bk)
    pkey = (kt, pk, rng.random(npr) >= null_p) if null_p else (kt, pk)
    bcols = [bkey] + [column(rng, nb, int(rng.choice([pl.INT32, pl.INT64, pl.FP64])), 0.1 if seed % 5 == 0 else 0.0)
                      for _ in range(1 + seed % 2)]
    pcols = [pkey, column(rng, npr, int(rng.choice([pl.INT32, pl.INT64])), 0.0)]
    bt, pt = pl.make_table(bcols), pl.make_table(pcols)
    p = pl.Plan()
    b = p.new_scan_node(0, [(i, c[0]) for i, c in enumerate(bcols)])
    s = p.new_scan_node(1, [(i, c[0]) for i, c in enumerate(pcols)])
    both = [c[0] for c in bcols] + [c[0] for c in pcols]
    build_left = bool(seed % 2 == 0)
    if build_left:
        j = p.new_join_node(True, b, s, 0, 0, [(i, t) for i, t in enumerate(both)])
    else:  # probe on the left, build on the right (the shape of every JOB hash join)
        both = [c[0] for c in pcols] + [c[0] for c in bcols]
        j = p.new_join_node(False, s, b, 0, 0, [(i, t) for i, t in enumerate(both)])
    p.new_input(bt)
    p.new_input(pt)
    p.root = j
    want = _oracle.execute(p)
    got = capi.execute(p, ctx)
    assert got.num_rows == want.num_rows
    assert want.num_rows > 0
    assert pl.table_digest(got) == pl.table_digest(want)


@pytest.mark.parametrize("seed", range(8))
def test_midscale_broadcast_join(ctx, seed):
    """Build sides of 1..4096 rows (not partitioned: k_join_bcast) against 1-4 M probe rows:
    duplicate build keys (up to the whole build side on ONE key), NULL keys on both sides,
    every key type, nullable and wide payloads, probe keys that mostly miss."""
    rng = np.random.default_rng(2000 + seed)
    kt = [pl.INT32, pl.INT64, pl.FP64, pl.INT32][seed % 4]
    nb = [1, 3, 64, 500, 2048, 4096, 4095, 4096][seed]
    npr = int(rng.integers(1_000_000, 4_000_000))
    domain = 1 if seed == 7 else max(1, int(nb * rng.uniform(0.3, 3.0)))  # seed 7: one key, 4096 copies
    bk = keys(rng, nb, domain, kt)
    # most probe keys miss unless the domain is tiny; keep the output below ~2 |probe|
    pdomain = domain if seed != 7 else 5000
    pk = keys(rng, npr, max(pdomain, 1) * (1 if nb * 2 >= domain and seed != 7 else 1), kt)
    if seed == 7:
        pk = keys(rng, npr, 5000, kt)  # 1/5000 of the probe rows hit the 4096 copies
    elif domain > 0 and nb / domain > 2:  # several copies per build key: thin the hits out
        pk = keys(rng, npr, domain * 8, kt)
    bkey = (kt, bk, rng.random(nb) >= 0.1) if seed % 2 else (kt, bk)
    pkey = (kt, pk, rng.random(npr) >= 0.05) if seed % 3 == 0 else (kt, pk)
    bcols = [bkey, column(rng, nb, pl.INT64 if seed % 2 else pl.INT32, 0.2 if seed % 4 == 2 else 0.0)]
    pcols = [pkey, column(rng, npr, pl.INT32, 0.0)] + ([column(rng, npr, pl.FP64, 0.1)] if seed % 2 else [])
    bt, pt = pl.make_table(bcols), pl.make_table(pcols)
    p = pl.Plan()
    b = p.new_scan_node(0, [(i, c[0]) for i, c in enumerate(bcols)])
    s = p.new_scan_node(1, [(i, c[0]) for i, c in enumerate(pcols)])
    both = [c[0] for c in pcols] + [c[0] for c in bcols]
    j = p.new_join_node(False, s, b, 0, 0, [(i, t) for i, t in enumerate(both)])
    p.new_input(bt)
    p.new_input(pt)
    p.root = j
    want = _oracle.execute(p)
    got = capi.execute(p, ctx)
    assert got.num_rows == want.num_rows
    assert pl.table_digest(got) == pl.table_digest(want)


@pytest.mark.parametrize("kt", [pl.INT64, pl.FP64])
@pytest.mark.parametrize("probe_cols", [0, 1, 2])
def test_64_bit_keys_at_14_radix_bits(kt, probe_cols):
    """64-bit keys + ONE build carry word at 14 forced radix bits (three-array LDS table; a tagged
    table for this 12-byte shape — slot = {tag, build index}, dense entry = {high hash word, carry} —
    was built and measured SLOWER, 1.50 -> 2.01 ms per 100 M probe tuples, and is not shipped).
    1 M build rows in 2^14 partitions: ~60 keys per table, duplicate build keys (up to 4 copies), NULL
    keys, probe keys that miss, a hot probe key."""
    rng = np.random.default_rng(4000 + probe_cols + (7 if kt == pl.FP64 else 0))
    c = capi.Context(radix_bits=14)
    try:
        nb, npr = 1_000_000, 2_000_000
        domain = 600_000
        bk = keys(rng, nb, domain, kt)
        pk = keys(rng, npr, domain + 200_000, kt, hot=0.05)
        bcols = [(kt, bk, rng.random(nb) >= 0.02), column(rng, nb, pl.INT32, 0.0)]
        pcols = [(kt, pk, rng.random(npr) >= 0.03)] + [column(rng, npr, [pl.INT32, pl.INT64][i % 2] if probe_cols == 1 else pl.INT32, 0.0)
                                                      for i in range(probe_cols)]
        bt, pt = pl.make_table(bcols), pl.make_table(pcols)
        p = pl.Plan()
        b = p.new_scan_node(0, [(i, x[0]) for i, x in enumerate(bcols)])
        s = p.new_scan_node(1, [(i, x[0]) for i, x in enumerate(pcols)])
        both = [x[0] for x in bcols] + [x[0] for x in pcols]
        j = p.new_join_node(True, b, s, 0, 0, [(i, t) for i, t in enumerate(both) if i != 2])
        p.new_input(bt)
        p.new_input(pt)
        p.root = j
        want = _oracle.execute(p)
        got = capi.execute(p, c)
        assert got.num_rows == want.num_rows and want.num_rows > npr
        assert pl.table_digest(got) == pl.table_digest(want)
    finally:
        capi.destroy_context(c)


@pytest.mark.parametrize("bits", [16, 20, 21, 27])
def test_blocked_pairs_between_the_passes(bits):
    """key + one INT32 payload (packed plan) at forced 16 / 20 / 21 / 27 (clamped to 21) radix bits: two and three
    plain-histogram passes, the pairs BETWEEN them in blocks of 256 keys + 256 carries (BlockedLoader: the next histogram
    reads the keys only), the last pass back to 8-byte pairs for the join.  Sizes that end inside a block and
    inside a tile, NULL keys, duplicates, probe keys that miss; XCD-aware placement forced on."""
    import os

    rng = np.random.default_rng(5000 + bits)
    old = os.environ.get("RJ_TUNE_XCD_MIN_ROWS")
    os.environ["RJ_TUNE_XCD_MIN_ROWS"] = "0"
    try:
        c = capi.Context(radix_bits=bits)
    finally:
        if old is None:
            del os.environ["RJ_TUNE_XCD_MIN_ROWS"]
        else:
            os.environ["RJ_TUNE_XCD_MIN_ROWS"] = old
    try:
        for nb, npr in ((1_000_003, 2_500_001), (70_001, 100), (255, 257)):
            dom = max(2, int(nb * 0.7))
            bt = pl.make_table([(pl.INT32, keys(rng, nb, dom, pl.INT32), rng.random(nb) >= 0.03), column(rng, nb, pl.INT32, 0.0)])
            pt = pl.make_table([(pl.INT32, keys(rng, npr, dom + dom // 4, pl.INT32, hot=0.05), rng.random(npr) >= 0.02), column(rng, npr, pl.INT32, 0.0)])
            p = pl.Plan()
            b = p.new_scan_node(0, [(0, pl.INT32), (1, pl.INT32)])
            s_ = p.new_scan_node(1, [(0, pl.INT32), (1, pl.INT32)])
            p.root = p.new_join_node(True, b, s_, 0, 0, [(0, pl.INT32), (1, pl.INT32), (3, pl.INT32)])
            p.new_input(bt)
            p.new_input(pt)
            want = _oracle.execute(p)
            got = capi.execute(p, c)
            assert got.num_rows == want.num_rows
            assert pl.table_digest(got) == pl.table_digest(want)
    finally:
        capi.destroy_context(c)
