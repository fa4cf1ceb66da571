#!/bin/bash
# (1) sharded world-1 rehearsal after the host stopped waiting behind the compute streams
# (2) bit plans of the 12-byte two-pass join at 1 B rows: 18 bits (9+9, the default) against 17 bits cut 9+8 and 8+9
# (3) multi-device host path (RJ_DEVICES=0,0) with one uploading thread per device
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_sharded.py tests/test_cpp_shim.py -m gpu -x -q > gpurun_out/r03_n_tests.log 2>&1 || { tail -30 gpurun_out/r03_n_tests.log; exit 1; }
tail -2 gpurun_out/r03_n_tests.log
RJ_BENCH_FORCE_DIST=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/r03_n_dist1.json 2> gpurun_out/r03_n_dist1.err || { tail -20 gpurun_out/r03_n_dist1.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_n_dist1.json").readlines()[-1])
print("sharded world-1: %.2f G/s %.2f ms" % (d["value"] / 1e9, d["ms_per_step"]), d.get("exchange_ms"), {k: round(v["ms_per_step"], 2) for k, v in d["roofline"]["kernels"].items()})
PY
scripts/ab_env.sh "--no-extras --no-cpu-baseline --steps 5 --warmup 2 --workload config3" RJ_X=0 "RJ_TUNE_RADIX_BITS=17 RJ_TUNE_P1_BITS=9" "RJ_TUNE_RADIX_BITS=17 RJ_TUNE_P1_BITS=8" RJ_X=0 "RJ_TUNE_RADIX_BITS=17 RJ_TUNE_P1_BITS=9" "RJ_TUNE_RADIX_BITS=17 RJ_TUNE_P1_BITS=8" > gpurun_out/r03_n_bit_plans_ab.log 2>&1
cat gpurun_out/r03_n_bit_plans_ab.log
