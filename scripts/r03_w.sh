#!/bin/bash
# tiles per workgroup in the FIRST pass (auto = 8 at 1 B rows): 4 / 8 / 16 / 32, both 1 B-row workloads, six processes each
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for W in config3 uniform1b; do
  echo "######## $W"
  scripts/ab_reps.sh 6 "--no-extras --no-cpu-baseline --no-verify --steps 4 --warmup 1 --workload $W" RJ_X=auto_8 RJ_TUNE_TPG1=4 RJ_TUNE_TPG1=16 RJ_TUNE_TPG1=32 | grep -A8 "^####"
done > gpurun_out/r03_w_first_pass_group_size_ab.log 2>&1
cat gpurun_out/r03_w_first_pass_group_size_ab.log
