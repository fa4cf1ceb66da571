#!/bin/bash
# what would a second dependent global round trip per digit and tile cost the scatters (the chunk-table lookup of a
# histogram-free first pass)?  -DRJ_PT_EXTRA_LOOKUP=1 against the shipped build, config 3, six processes each
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
scripts/ab_reps.sh 6 "--no-extras --no-cpu-baseline --no-verify --steps 4 --warmup 1 --workload config3" RJ_X=shipped RJ_LIB_PATH=$PWD/radix-join_amd/librj_lookup.so | grep -A4 "^####" > gpurun_out/r03_ak_extra_lookup_ab.log 2>&1
cat gpurun_out/r03_ak_extra_lookup_ab.log
