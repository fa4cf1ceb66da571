#!/bin/bash
# round-3 check run: wide-carry tests, default bench (short), sharded world-1 rehearsal, RCCL probe both ways
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_wide_carry.py -x -q > gpurun_out/r03_e_tests.log 2>&1; tail -3 gpurun_out/r03_e_tests.log
timeout -k 10 500 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_e_bench.json 2> gpurun_out/r03_e_bench.err || exit 1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_e_bench.json").readlines()[-1])
print("config3 %.2f G/s %.2f ms" % (d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms_per_step"], 2) for k, v in d["roofline"]["kernels"].items()})
print("probe_read_frac", d["roofline"].get("probe_read_frac"))
for k, v in d.get("configs", {}).items():
    print(k, "%.2f G/s %.2f ms" % (v["value"] / 1e9, v["ms_per_step"]), {a: round(b, 2) for a, b in v["kernels_ms_per_step"].items()})
print("plan_ms", {k: (round(v["ms"], 2), round(v["ms_first_call"], 2)) if isinstance(v, dict) else round(v, 2) for k, v in d.get("plan_ms", {}).items()})
PY
RJ_BENCH_FORCE_DIST=1 RJ_DIAG=2 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 3 --warmup 1 > gpurun_out/r03_e_dist1.json 2> gpurun_out/r03_e_dist1.err || { tail -20 gpurun_out/r03_e_dist1.err; exit 1; }
grep "rj sharded\|rj comm" gpurun_out/r03_e_dist1.err | tail -12
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_e_dist1.json").readlines()[-1])
print("sharded world-1: %.2f G/s %.2f ms" % (d["value"] / 1e9, d["ms_per_step"]), d.get("exchange_ms"), {k: round(v["ms_per_step"], 2) for k, v in d["roofline"]["kernels"].items()})
PY
echo "== probe, default load order" > gpurun_out/r03_f_rccl_probe.log
RJ_DIAG=2 RJ_EXCHANGE_TIMEOUT_MS=20000 timeout -k 5 120 python scripts/rccl_probe.py both >> gpurun_out/r03_f_rccl_probe.log 2>&1
echo "exit $?" >> gpurun_out/r03_f_rccl_probe.log
echo "== probe, ROCm's librccl forced by path into a process that runs on PyTorch's HIP runtime" >> gpurun_out/r03_f_rccl_probe.log
RJ_RCCL_PATH=/opt/rocm/lib/librccl.so.1 RJ_DIAG=2 RJ_EXCHANGE_TIMEOUT_MS=20000 timeout -k 5 120 python scripts/rccl_probe.py both >> gpurun_out/r03_f_rccl_probe.log 2>&1
echo "exit $?" >> gpurun_out/r03_f_rccl_probe.log
grep -v "^\[W\|amdgpu.ids" gpurun_out/r03_f_rccl_probe.log | tail -30
