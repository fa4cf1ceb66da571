#!/bin/bash
# probe tuples per thread and output reservation of the join (JN_SPT 8 = shipped, 16), six processes each
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
P=$PWD/radix-join_amd
for W in config3 uniform1b; do
  echo "######## $W"
  scripts/ab_reps.sh 6 "--no-extras --no-cpu-baseline --no-verify --steps 4 --warmup 1 --workload $W" RJ_X=shipped_spt8 RJ_LIB_PATH=$P/librj_spt16.so | grep -A4 "^####"
done > gpurun_out/r03_ai_join_spt_ab.log 2>&1
cat gpurun_out/r03_ai_join_spt_ab.log
