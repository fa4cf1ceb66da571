#!/usr/bin/env python3
"""rj_table_from_csv at scale: ~0.6 GB of cast_info-shaped CSV text (id, person_id, movie_id, note with
quotes and NULLs, nr_order, a double), filtered, against the oracle's restatement of the reference path
(pages byte for byte) — a robustness check of the two-level segment scans and the page walk far above
the test suite's sizes.    python scripts/ingest_big.py [rows]   (on the GPU box; default 12 M rows)"""
import io
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [os.path.join(ROOT, "radix-join_amd"), os.path.join(ROOT, "tests")]
import _oracle  # noqa: E402
from pyrj import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12_000_000
rng = np.random.default_rng(5)
# notes as they look in the harness's dialect: quotes around anything with a comma or a quote, a quote inside
# quotes behind a backslash, a backslash outside quotes is just a character; an empty field is NULL
notes = ['(voice)', '(uncredited)', '"(as \\"Himself\\")"', '"a, b"', '', '(archive footage)', 'x\\y', '"(credit only)"']
t0 = time.time()
person = rng.integers(1, 4_000_000, n)
movie = rng.integers(1, 2_500_000, n)
note = rng.integers(0, len(notes), n)
order = np.where(rng.random(n) < 0.4, -1, rng.integers(1, 100, n))
score = np.round(rng.random(n) * 1000, 3)
parts = []
CH = 1 << 20
for a in range(0, n, CH):
    b = min(n, a + CH)
    parts.append("".join(f"{i + 1},{person[i]},{movie[i]},{notes[note[i]]},{'' if order[i] < 0 else order[i]},{score[i]}\n"
                         for i in range(a, b)).encode())
text = b"".join(parts)
del parts
print(f"{n} rows, {len(text) / 1e6:.0f} MB of text (written in {time.time() - t0:.0f} s)", flush=True)
types = [1, 0, 0, 3, 0, 2]  # INT64, INT32, INT32, VARCHAR, INT32, FP64
filt = [("GT", 2, 1_000_000), ("IS_NOT_NULL", 4), ("AND",), ("LIKE", 3, b"(%)"), ("OR",)]
ctx = capi.Context(profile=2 if os.environ.get("RJ_INGEST_PROFILE") == "1" else False)
for k in range(2):
    t0 = time.time()
    t = ctx.from_csv(text, types, filt)
    dt = time.time() - t0
    print(f"device: {dt * 1e3:.0f} ms = {len(text) / dt / 1e9:.2f} GB/s of text", flush=True)
    if k == 0:
        t.release()
        if os.environ.get("RJ_INGEST_PROFILE") == "1":
            ctx.profile_reset()
if os.environ.get("RJ_INGEST_PROFILE") == "1":
    print("   " + "  ".join("%s=%.2f(%d)" % (k["name"], k["total_ms"], k["launches"]) for k in ctx.profile()), flush=True)
got = ctx.table_to_host(t)
t.release()
t0 = time.time()
want = _oracle.from_csv(text, types, filt)
print(f"oracle: {time.time() - t0:.1f} s; rows kept {want.num_rows}", flush=True)
assert got.num_rows == want.num_rows, (got.num_rows, want.num_rows)
for c, (a, b) in enumerate(zip(got.columns, want.columns)):
    assert a.pages.shape == b.pages.shape and np.array_equal(a.pages, b.pages), f"column {c} differs"
print("pages identical to the oracle's, all", len(types), "columns")
capi.destroy_context(ctx)
