"""The N>1 path on CPU: world_size 2 over gloo.  The exchange logic of pyrj.dist.ShardedJoin
(count all-to-all, split sizes, tuple all-to-all) runs for real; the two local stages are CPU
stand-ins (numpy partition with the library's hash + the oracle's join)."""
import os
import pickle
import socket
import subprocess
import sys

import numpy as np

from pyrj import hashing

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_join_world2_gloo(tmp_path):
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
        os.path.join(HERE, "_dist_worker.py"), str(tmp_path),
    ]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    parts = [pickle.load(open(tmp_path / f"rank{k}.pkl", "rb")) for k in range(2)]
    # every rank received exactly the keys it owns (top hash bit), from BOTH shards
    for k, p in enumerate(parts):
        assert (hashing.owner_rank(p["build_keys"], 2) == k).all()
        assert (hashing.owner_rank(p["probe_keys"], 2) == k).all()
    all_b = np.concatenate([p["shard"][0][p["shard"][1]] for p in parts])
    all_p = np.concatenate([p["shard"][2][p["shard"][3]] for p in parts])
    assert sum(len(p["build_keys"]) for p in parts) == len(all_b)
    assert sum(len(p["probe_keys"]) for p in parts) == len(all_p)
    # the union of the ranks' results is the join of the union of the shards
    brows, prows = [], []
    for p in parts:
        bk, bv, pk, pv, rank, nb, npr = p["shard"]
        brows += [(int(k), int(i + 100000 * rank)) for i, (k, v) in enumerate(zip(bk, bv)) if v]
        prows += [(int(k), int(i + 100000 * rank)) for i, (k, v) in enumerate(zip(pk, pv)) if v]
    by_key = {}
    for k, pay in brows:
        by_key.setdefault(k, []).append(pay)
    expect = sorted((k, b, pay) for k, pay in prows for b in by_key.get(k, []))
    got = sorted(tuple(r) for p in parts for r in p["rows"])
    assert got == expect


def test_hash_roundtrip_and_ownership_bits():
    rng = np.random.default_rng(0)
    k = rng.integers(-(2**31), 2**31 - 1, 100000).astype(np.int32)
    h = hashing.fmix32(k.view(np.uint32))
    assert np.array_equal(hashing.unfmix32(h).view(np.int32), k)
    assert len(np.unique(h)) == len(np.unique(k))  # bijection
    for n in (1, 2, 4, 8):
        o = hashing.owner_rank(k, n)
        assert o.min() >= 0 and o.max() < n
        if n > 1:
            assert np.bincount(o, minlength=n).min() > 100000 / n * 0.9
