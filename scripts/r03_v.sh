#!/bin/bash
# probe tuples per heavy task (JN_HEAVY): 16 K / 32 K (shipped) / 64 K / 128 K, config 3 (Zipf-0.9 probe side)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
P=$PWD/radix-join_amd
scripts/ab_reps.sh 6 "--no-extras --no-cpu-baseline --no-verify --steps 4 --warmup 1 --workload config3" RJ_X=shipped_32k RJ_LIB_PATH=$P/librj_h16.so RJ_LIB_PATH=$P/librj_h64.so RJ_LIB_PATH=$P/librj_h128.so > gpurun_out/r03_v_heavy_task_size_ab.log 2>&1
tail -6 gpurun_out/r03_v_heavy_task_size_ab.log
