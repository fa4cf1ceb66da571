#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python scripts/first_call_diag.py 13d > gpurun_out/r03_l_first_call_13d.log 2>&1
grep "rj host\] execute\|call \|build_context" gpurun_out/r03_l_first_call_13d.log
scripts/ab_env.sh "--no-extras --no-cpu-baseline --steps 5 --warmup 2 --workload config3" RJ_X=0 RJ_TUNE_TPG2=2 RJ_X=0 RJ_TUNE_TPG2=2 RJ_TUNE_TPG2=3 RJ_TUNE_TPG2=4 RJ_X=0 RJ_TUNE_TPG2=2 > gpurun_out/r03_l_tpg2_ab.log 2>&1
scripts/ab_env.sh "--no-extras --no-cpu-baseline --steps 5 --warmup 2 --workload uniform1b" RJ_X=0 RJ_TUNE_TPG2=2 RJ_X=0 RJ_TUNE_TPG2=4 >> gpurun_out/r03_l_tpg2_ab.log 2>&1
cat gpurun_out/r03_l_tpg2_ab.log
