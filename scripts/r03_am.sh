#!/bin/bash
# what would a hot-key test per tuple (one probe of a 4096-entry LDS set) cost the first pass?  -DRJ_PT_HOT_PROBE=1
# (the probe sits in every histogram / scatter launch: only the pass-1 columns are what a skew bypass would pay)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
scripts/ab_reps.sh 6 "--no-extras --no-cpu-baseline --no-verify --steps 4 --warmup 1 --workload config3" RJ_X=shipped RJ_LIB_PATH=$PWD/radix-join_amd/librj_hot.so | grep -A4 "^####" > gpurun_out/r03_am_hot_probe_ab.log 2>&1
cat gpurun_out/r03_am_hot_probe_ab.log
