"""Sharded (multi-GPU) join: one process per GPU, one all-to-all.

The join shards by hash of the key: rank r owns the tuples whose top hash digit is
r.  The only data-path collective is the radix re-distribution step:

    stage A (local)   rj_shard_partition: decode + hash + partition by rank
    exchange          all-to-all of per-destination counts, then of key and carry
                      arrays (RCCL over xGMI through torch.distributed; gloo on CPU
                      in the tests)
    stage B (local)   rj_join_tuples: radix passes + build/probe on what arrived

There is no reference counterpart (the reference is a single CPU process,
SURVEY.md §2a); semantics are those of one JoinNode(build_left=true,
out={key, build payload, probe payload}) over the union of all shards.

``ShardedJoin`` only moves opaque tensors, so it runs unchanged over the HIP ops
(``GpuOps``) on a GPU box and over a CPU stand-in in the world_size-2 gloo tests.
"""
from __future__ import annotations

import numpy as np


class GpuOps:
    """Stage A / B on one MI355X through the C-ABI; buffers are torch CUDA tensors."""

    def __init__(self, ctx, device=None):
        import torch

        self.torch = torch
        self.ctx = ctx
        self.device = device or torch.device("cuda", torch.cuda.current_device())

    def partition(self, table, n_rows, n_ranks, key_col=0, carry_col=1):
        t = self.torch
        key = t.empty(max(n_rows, 1), dtype=t.int32, device=self.device)
        carry = t.empty(max(n_rows, 1), dtype=t.int32, device=self.device)
        t.cuda.current_stream().synchronize()
        n, counts = self.ctx.shard_partition(table, key_col, carry_col, n_ranks, key.data_ptr(), carry.data_ptr())
        return key[:n], carry[:n], counts

    def empty(self, n):
        return self.torch.empty(max(n, 1), dtype=self.torch.int32, device=self.device)[:n]

    def join(self, bkey, bcarry, pkey, pcarry, skip_rank_bits):
        self.torch.cuda.current_stream().synchronize()
        return self.ctx.join_tuples(
            (bkey.numel(), bkey.data_ptr(), bcarry.data_ptr()),
            (pkey.numel(), pkey.data_ptr(), pcarry.data_ptr()),
            skip_rank_bits=skip_rank_bits,
            hashed=True,
        )


class ShardedJoin:
    """The exchange logic; ``ops`` provides partition/empty/join on its device."""

    def __init__(self, ops, group=None):
        import torch
        import torch.distributed as dist

        self.torch, self.dist, self.ops, self.group = torch, dist, ops, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.rank_bits = max(0, (self.world - 1).bit_length())
        if 1 << self.rank_bits != self.world:
            raise ValueError("world size must be a power of two")

    def _exchange(self, key, carry, counts):
        t, d = self.torch, self.dist
        dev = key.device
        # RCCL moves device tensors directly over xGMI; gloo (CPU tests, and the one-GPU
        # rehearsal of the multi-rank bench) needs host tensors, so stage through the host
        host = d.get_backend(self.group) == "gloo" and dev.type != "cpu"
        cdev = t.device("cpu") if host else dev
        send = t.tensor(counts, dtype=t.int64, device=cdev)
        recv = t.empty_like(send)
        d.all_to_all_single(recv, send, group=self.group)
        rc = [int(x) for x in recv.tolist()]
        n = sum(rc)
        if host:
            rkey_h, rcarry_h = t.empty(n, dtype=key.dtype), t.empty(n, dtype=carry.dtype)
            d.all_to_all_single(rkey_h, key.cpu(), output_split_sizes=rc, input_split_sizes=list(counts), group=self.group)
            d.all_to_all_single(rcarry_h, carry.cpu(), output_split_sizes=rc, input_split_sizes=list(counts), group=self.group)
            rkey, rcarry = self.ops.empty(n), self.ops.empty(n)
            rkey.copy_(rkey_h)
            rcarry.copy_(rcarry_h)
            return rkey, rcarry
        rkey, rcarry = self.ops.empty(n), self.ops.empty(n)
        d.all_to_all_single(rkey, key, output_split_sizes=rc, input_split_sizes=list(counts), group=self.group)
        d.all_to_all_single(rcarry, carry, output_split_sizes=rc, input_split_sizes=list(counts), group=self.group)
        return rkey, rcarry

    def run(self, build_table, build_rows, probe_table, probe_rows):
        """-> this rank's slice of the join result (``ops.join`` return value)."""
        bk, bc, bcnt = self.ops.partition(build_table, build_rows, self.world)
        pk, pc, pcnt = self.ops.partition(probe_table, probe_rows, self.world)
        if self.world > 1:
            bk, bc = self._exchange(bk, bc, bcnt)
            pk, pc = self._exchange(pk, pc, pcnt)
        return self.ops.join(bk, bc, pk, pc, self.rank_bits)


def split_rows(arrs, n_ranks):
    """Contiguous row shards of parallel numpy arrays."""
    n = arrs[0].shape[0]
    cuts = [n * r // n_ranks for r in range(n_ranks + 1)]
    return [[a[cuts[r] : cuts[r + 1]] for a in arrs] for r in range(n_ranks)]


def virtual_rank_join(ctx, build_table, probe_table, n_ranks):
    """Run the n-rank sharded path on ONE GPU: every virtual rank runs stage A on its
    shard, the all-to-all is done by slicing, every virtual rank runs stage B.
    Inputs are (key INT32, payload INT32) tables; returns the per-rank result tables."""
    import torch

    from . import pages as pg
    from . import plan as pl

    ops = GpuOps(ctx)

    def shards(t):
        k, kv = pg.unpack_fixed(t.columns[0].pages, t.num_rows, pl.INT32)
        v, _ = pg.unpack_fixed(t.columns[1].pages, t.num_rows, pl.INT32)
        out = []
        for ks, vs, kvs in split_rows([k, v, kv], n_ranks):
            st = pl.make_table([(pl.INT32, ks, kvs), (pl.INT32, vs)])
            out.append((ctx.upload(st), st.num_rows))
        return out

    sent = {"b": [], "p": []}
    for side, table in (("b", build_table), ("p", probe_table)):
        for tbl, rows in shards(table):
            k, c, counts = ops.partition(tbl, rows, n_ranks)
            offs = np.concatenate([[0], np.cumsum(counts)])
            sent[side].append([(k[offs[r] : offs[r + 1]].clone(), c[offs[r] : offs[r + 1]].clone()) for r in range(n_ranks)])
            tbl.release()
    results = []
    rank_bits = (n_ranks - 1).bit_length()
    for r in range(n_ranks):
        bk = torch.cat([sent["b"][s][r][0] for s in range(n_ranks)])
        bc = torch.cat([sent["b"][s][r][1] for s in range(n_ranks)])
        pk = torch.cat([sent["p"][s][r][0] for s in range(n_ranks)])
        pc = torch.cat([sent["p"][s][r][1] for s in range(n_ranks)])
        torch.cuda.synchronize()
        res = ops.join(bk, bc, pk, pc, rank_bits)
        results.append(res.to_table())
        res.free()
    return results
