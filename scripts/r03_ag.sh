#!/bin/bash
# the multi-GPU driver of bench.py at world size 1 over RCCL (one-GPU box), on the round's final code
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
RJ_BENCH_FORCE_DIST=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/r03_ag_dist1.json 2> gpurun_out/r03_ag_dist1.err || { tail -20 gpurun_out/r03_ag_dist1.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_ag_dist1.json").readlines()[-1])
print("sharded world-1: %.2f G/s %.2f ms" % (d["value"] / 1e9, d["ms_per_step"]), d.get("exchange_ms"), {k: round(v["ms_per_step"], 2) for k, v in d["roofline"]["kernels"].items()}, d["config"]["verified"])
PY
