#!/bin/bash
# clean sharded world-1 rehearsal (no diagnostics), folded and unfolded; scatter phase stamps; f2 survey;
# probe with a truncated communicator id
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
RJ_BENCH_FORCE_DIST=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/r03_g_dist1.json 2> gpurun_out/r03_g_dist1.err || { tail -20 gpurun_out/r03_g_dist1.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_g_dist1.json").readlines()[-1])
print("sharded world-1: %.2f G/s %.2f ms" % (d["value"] / 1e9, d["ms_per_step"]), d.get("exchange_ms"), {k: round(v["ms_per_step"], 2) for k, v in d["roofline"]["kernels"].items()})
PY
RJ_TUNE_FOLD_OWNER=0 RJ_BENCH_FORCE_DIST=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/r03_g_dist1_unfolded.json 2> gpurun_out/r03_g_dist1_unfolded.err || { tail -20 gpurun_out/r03_g_dist1_unfolded.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_g_dist1_unfolded.json").readlines()[-1])
print("sharded world-1, owner digit NOT folded: %.2f G/s %.2f ms" % (d["value"] / 1e9, d["ms_per_step"]), d.get("exchange_ms"), {k: round(v["ms_per_step"], 2) for k, v in d["roofline"]["kernels"].items()})
PY
echo "== scatter phase stamps (diagnostic build)" > gpurun_out/r03_g_ptdiag.log
RJ_LIB_PATH=$PWD/radix-join_amd/librj_ptdiag.so RJ_DIAG=3 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-verify 2>&1 | grep "scatter pass" | tail -8 >> gpurun_out/r03_g_ptdiag.log
RJ_LIB_PATH=$PWD/radix-join_amd/librj_ptdiag.so RJ_DIAG=3 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-verify --workload uniform1b 2>&1 | grep "scatter pass" | tail -8 >> gpurun_out/r03_g_ptdiag.log
cat gpurun_out/r03_g_ptdiag.log
timeout -k 10 500 python scripts/f2_survey.py > gpurun_out/r03_g_f2_survey.log 2> gpurun_out/r03_g_f2_survey.err; tail -5 gpurun_out/r03_g_f2_survey.log; tail -3 gpurun_out/r03_g_f2_survey.err
echo "== probe, communicator id cut at its first NUL byte" > gpurun_out/r03_g_rccl_probe_truncated.log
RJ_DIAG=2 RJ_EXCHANGE_TIMEOUT_MS=15000 timeout -k 5 90 python scripts/rccl_probe.py truncated >> gpurun_out/r03_g_rccl_probe_truncated.log 2>&1
echo "exit $?" >> gpurun_out/r03_g_rccl_probe_truncated.log
grep -v "^\[W\|amdgpu.ids" gpurun_out/r03_g_rccl_probe_truncated.log | tail -12
