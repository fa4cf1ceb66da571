#!/bin/bash
# final records of the round: full GPU suite, rocprofv3 + PMC summaries of the three workloads, default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_u_tests.log 2>&1; tail -3 gpurun_out/r03_u_tests.log
rm -f gpurun_out/traffic.json
scripts/profile_run.sh r03_zz_config3_final config3 && scripts/profile_run.sh r03_zz_uniform1b_final uniform1b && scripts/profile_run.sh r03_zz_config2_final config2 || { echo "profile run failed"; tail -5 gpurun_out/prof_*/trace.log; exit 1; }
timeout -k 10 900 python bench.py > gpurun_out/r03_u_bench.json 2> gpurun_out/r03_u_bench.err; tail -c 300 gpurun_out/r03_u_bench.json
