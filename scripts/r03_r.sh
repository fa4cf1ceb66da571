#!/bin/bash
# 18 radix bits cut 8+10 and 10+8 (kernels can fan out 1024 ways) against 9+9, both 1 B-row workloads
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
scripts/ab_env.sh "--no-extras --no-cpu-baseline --steps 5 --warmup 2 --workload config3" RJ_X=0 RJ_TUNE_P1_BITS=8 RJ_TUNE_P1_BITS=10 RJ_X=0 RJ_TUNE_P1_BITS=8 RJ_TUNE_P1_BITS=10 > gpurun_out/r03_r_bits_8_10_ab.log 2>&1
scripts/ab_env.sh "--no-extras --no-cpu-baseline --steps 5 --warmup 2 --workload uniform1b" RJ_X=0 RJ_TUNE_P1_BITS=8 RJ_TUNE_P1_BITS=10 RJ_X=0 RJ_TUNE_P1_BITS=8 >> gpurun_out/r03_r_bits_8_10_ab.log 2>&1
cat gpurun_out/r03_r_bits_8_10_ab.log
