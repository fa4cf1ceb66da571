#!/usr/bin/env python3
"""Diagnostic: does RCCL come up on this box, first through torch.distributed, then through
librj's own dlopen'ed communicator (world size 1)?  usage: rccl_probe.py [torch|rj|both]"""
import faulthandler
import os
import sys

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radix-join_amd")]
what = sys.argv[1] if len(sys.argv) > 1 else "both"
# "truncated": as "rj", but the communicator id is cut at its first NUL byte and zero-padded — what
# reading a ctypes c_char array as `.bytes` / `.value` does to an ncclUniqueId
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
if what in ("torch", "both"):
    import torch
    import torch.distributed as dist

    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("torch nccl all_reduce ok", t.tolist(), flush=True)
if what in ("rj", "both", "truncated"):
    from pyrj import capi

    cid = capi.make_comm_id()
    print("comm id ok", len(cid), "first NUL at byte", cid.find(b"\0"), flush=True)
    if what == "truncated":
        cut = cid.find(b"\0")
        cid = cid[:cut] + b"\0" * (128 - cut)
    with open("/proc/self/maps") as f:
        libs = sorted({ln.split()[-1] for ln in f if "rccl" in ln or "amdhip64" in ln})
    print("loaded:", libs, flush=True)
    import time

    t0 = time.time()
    try:
        ctx = capi.Context(devices=[0], world_size=1, rank_base=0, comm_id=cid, exchange=capi.EXCHANGE_RCCL)
    except capi.RjError as e:
        # (bring-up is bounded by RJ_EXCHANGE_TIMEOUT_MS; a helper thread may still sit inside RCCL)
        print(f"rj context FAILED after {time.time() - t0:.1f} s: {e}", flush=True)
        with open("/proc/self/maps") as f:
            libs = sorted({ln.split()[-1] for ln in f if "rccl" in ln or "amdhip64" in ln})
        print("loaded at failure:", libs, flush=True)
        os._exit(3)
    print(f"rj context with RCCL communicator ok after {time.time() - t0:.1f} s", flush=True)
    with open("/proc/self/maps") as f:
        libs = sorted({ln.split()[-1] for ln in f if "rccl" in ln or "amdhip64" in ln})
    print("loaded:", libs, flush=True)
    ctx.destroy()
    print("destroyed", flush=True)
