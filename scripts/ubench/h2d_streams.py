#!/usr/bin/env python3
"""Host -> device copy rate from pinned memory in 32 MiB chunks: one stream against two and four
(does a second copy engine add to what one hipMemcpyAsync stream moves over the PCIe link?)."""
import time

import torch

assert torch.cuda.is_available()
dev = torch.device("cuda:0")
total, chunk = 4 << 30, 32 << 20
src = torch.empty(total, dtype=torch.uint8).pin_memory()
src.fill_(7)
dst = torch.empty(total, dtype=torch.uint8, device=dev)
for ns in (1, 2, 4, 1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k, o in enumerate(range(0, total, chunk)):
        with torch.cuda.stream(streams[k % ns]):
            dst[o:o + chunk].copy_(src[o:o + chunk], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{ns} stream(s): {total / dt / 1e9:.1f} GB/s", flush=True)
