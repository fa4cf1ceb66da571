"""gloo worker for tests/test_exchange_plan.py: the exchange step of a sharded join rehearsed
across real processes on CPU.  Every rank stands in for stage A with numpy (tuples of its shard
grouped by (owner rank, first local digit), owner-major, with the library's hash), all-gathers
its counts, asks the LIBRARY for its half of the exchange (rj_exchange_plan — the same host
function ShardedExec runs between the count all-gather and the all-to-all), moves the tuples with
torch.distributed.all_to_all_single using exactly those offsets and counts, and checks that what
arrived is what the plan says: run (digit k, source s) sits at [seg_begin, seg_end) of index
k * world + s and holds precisely source s's tuples of digit k for this rank."""
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "radix-join_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from pyrj import capi, hashing  # noqa: E402


def main():
    out_dir, sbits = sys.argv[1], int(sys.argv[2])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rb, S = world.bit_length() - 1, 1 << sbits
    rng = np.random.default_rng(500 + rank)
    n = 20000 + 3500 * rank  # ragged shards
    keys = rng.integers(0, 1 << 20, n).astype(np.int32)
    if rank == 1:
        keys[: n // 3] = 12345  # a hot key: one (owner, digit) run is much longer than the others
    h = hashing.fmix32(keys.view(np.uint32))
    owner = (h >> np.uint32(32 - rb)).astype(np.int64) if rb else np.zeros(n, np.int64)
    sub = (h & np.uint32(S - 1)).astype(np.int64)
    digit = owner * S + sub  # stage A's composite digit
    order = np.argsort(digit, kind="stable")
    send = np.stack([h[order].astype(np.int64), np.full(n, rank, np.int64), np.arange(n, dtype=np.int64)[order]], axis=1)
    mine = np.bincount(digit, minlength=world * S).astype(np.int64)
    # count all-gather
    cnt = [torch.zeros(world * S, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(cnt, torch.from_numpy(mine))
    cnt = torch.stack(cnt).numpy().reshape(world, world, S)
    plan = capi.exchange_plan(world, S, rank, cnt)
    # the all-to-all, with the plan's offsets (one contiguous slice per peer, source-major receive)
    assert np.array_equal(plan["send_off"], np.concatenate([[0], np.cumsum(plan["send_cnt"])[:-1]]))
    assert np.array_equal(plan["recv_off"], np.concatenate([[0], np.cumsum(plan["recv_cnt"])[:-1]]))
    recv = torch.zeros((plan["n_recv"], 3), dtype=torch.int64)
    dist.all_to_all_single(recv, torch.from_numpy(send), [int(c) for c in plan["recv_cnt"]], [int(c) for c in plan["send_cnt"]])
    recv = recv.numpy()
    # what arrived, run by run
    seen = np.zeros(plan["n_recv"], bool)
    for k in range(S):
        for s in range(world):
            b, e = int(plan["seg_begin"][k * world + s]), int(plan["seg_end"][k * world + s])
            assert e - b == cnt[s, rank, k]
            run = recv[b:e]
            assert (run[:, 1] == s).all()
            hh = run[:, 0].astype(np.uint32)
            assert ((hh & np.uint32(S - 1)) == k).all()
            if rb:
                assert ((hh >> np.uint32(32 - rb)) == rank).all()
            assert not seen[b:e].any()
            seen[b:e] = True
        assert plan["part_off"][k + 1] - plan["part_off"][k] == cnt[:, rank, k].sum()
    assert seen.all()
    pickle.dump({"n_sent": n, "n_recv": plan["n_recv"], "rows": recv[:, 1:].copy()}, open(os.path.join(out_dir, f"rank{rank}.pkl"), "wb"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
