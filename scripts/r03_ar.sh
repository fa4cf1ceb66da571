#!/bin/bash
# the round's final code, ten processes per workload (every process draws its own scratch placement): mean and spread
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for W in config3 uniform1b config2; do
  echo "######## $W"
  scripts/ab_reps.sh 10 "--no-extras --no-cpu-baseline --no-verify --steps 4 --warmup 1 --workload $W" RJ_X=final | grep -A3 "^####"
done > gpurun_out/r03_ar_final_reps.log 2>&1
cat gpurun_out/r03_ar_final_reps.log
