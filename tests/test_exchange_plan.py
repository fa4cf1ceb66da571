"""The exchange layout of a sharded join (csrc/rj_xplan.cpp, C-ABI rj_exchange_plan): host
arithmetic over the all-gathered count tensor, checked without a GPU —
  * for worlds of 2 / 4 / 8 ranks and 1 .. 64 local digits: every pair of ranks agrees on what
    moves between them, send slices and receive ranges tile their buffers, the arriving runs are
    listed digit-major and add up to the first-level partitions;
  * the "more than 2^32 tuples on one rank" refusal is the same on every rank;
  * and across REAL processes (gloo, world 2 and 4): tuples moved with exactly the plan's offsets
    land where the plan says (tests/_xplan_worker.py)."""
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest

from pyrj import capi

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("subs", [1, 2, 64])
def test_every_pair_of_ranks_agrees(world, subs):
    rng = np.random.default_rng(world * 100 + subs)
    cnt = rng.integers(0, 5000, (world, world, subs)).astype(np.uint64)
    cnt[rng.random(cnt.shape) < 0.15] = 0  # empty runs
    if world > 1:
        cnt[1, :, :] = 0  # a rank with an empty shard
    plans = [capi.exchange_plan(world, subs, r, cnt) for r in range(world)]
    for a in range(world):
        pa = plans[a]
        # my stage-A output is owner-major: slices one behind the other, in rank order
        assert np.array_equal(pa["send_cnt"], cnt[a].sum(axis=1))
        assert np.array_equal(pa["send_off"], np.concatenate([[0], np.cumsum(pa["send_cnt"])[:-1]]))
        assert np.array_equal(pa["recv_cnt"], cnt[:, a].sum(axis=1))
        assert np.array_equal(pa["recv_off"], np.concatenate([[0], np.cumsum(pa["recv_cnt"])[:-1]]))
        assert pa["n_recv"] == int(cnt[:, a].sum())
        for b in range(world):
            assert pa["send_cnt"][b] == plans[b]["recv_cnt"][a]  # pairwise agreement
        # the arriving runs: digit-major list, each inside its source's range, digits in order
        seen = np.zeros(pa["n_recv"], bool)
        for s in range(world):
            pos = int(pa["recv_off"][s])
            for k in range(subs):
                b0, e0 = int(pa["seg_begin"][k * world + s]), int(pa["seg_end"][k * world + s])
                assert b0 == pos and e0 - b0 == cnt[s, a, k]
                seen[b0:e0] = True
                pos = e0
            assert pos == pa["recv_off"][s] + pa["recv_cnt"][s]
        assert seen.all()
        assert np.array_equal(np.diff(pa["part_off"].astype(np.int64)), cnt[:, a].sum(axis=0).astype(np.int64))


def test_overflow_is_refused_alike_on_every_rank():
    world, subs = 4, 8
    cnt = np.full((world, world, subs), 1000, dtype=np.uint64)
    cnt[:, 2, :] = 2**32 // (world * subs) + 7  # rank 2 would receive just over 2^32 tuples
    msgs = []
    for r in range(world):
        with pytest.raises(capi.RjError) as e:
            capi.exchange_plan(world, subs, r, cnt)
        assert e.value.code == 5  # RJ_ERR_UNSUPPORTED
        msgs.append(e.value.message)
    assert len(set(msgs)) == 1 and "rank 2" in msgs[0]
    with pytest.raises(capi.RjError) as e:
        capi.exchange_plan(4, 8, 4, np.zeros((4, 4, 8), np.uint64))  # rank out of range
    assert e.value.code == 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,sbits", [(2, 3), (4, 0), (4, 2)])
def test_exchange_over_gloo_lands_where_the_plan_says(tmp_path, world, sbits):
    cmd = [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
        os.path.join(HERE, "_xplan_worker.py"), str(tmp_path), str(sbits),
    ]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    parts = [pickle.load(open(tmp_path / f"rank{k}.pkl", "rb")) for k in range(world)]
    assert sum(p["n_sent"] for p in parts) == sum(p["n_recv"] for p in parts)
    # every tuple of every shard arrived exactly once somewhere
    rows = np.concatenate([p["rows"] for p in parts])
    assert len(np.unique(rows, axis=0)) == len(rows) == sum(p["n_sent"] for p in parts)
