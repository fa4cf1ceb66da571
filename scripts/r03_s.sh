#!/bin/bash
# does the 1024-digit build change the 9-bit kernels?  shipped library against the cap-10 build, interleaved,
# with the 17-bit 8+9 plan (both can run it) and 18 bits cut 8+10 (cap-10 only)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
A="--no-extras --no-cpu-baseline --steps 5 --warmup 2 --workload config3"
C10="RJ_LIB_PATH=$PWD/radix-join_amd/librj_cap10.so"
scripts/ab_env.sh "$A" RJ_X=shipped "$C10" "RJ_TUNE_RADIX_BITS=17 RJ_TUNE_P1_BITS=8" "$C10 RJ_TUNE_RADIX_BITS=17 RJ_TUNE_P1_BITS=8" "$C10 RJ_TUNE_P1_BITS=8" \
   RJ_X=shipped "$C10" "RJ_TUNE_RADIX_BITS=17 RJ_TUNE_P1_BITS=8" "$C10 RJ_TUNE_RADIX_BITS=17 RJ_TUNE_P1_BITS=8" "$C10 RJ_TUNE_P1_BITS=8" \
   RJ_X=shipped "$C10" "$C10 RJ_TUNE_P1_BITS=8" > gpurun_out/r03_s_cap10_ab.log 2>&1
cat gpurun_out/r03_s_cap10_ab.log
