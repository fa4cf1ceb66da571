#!/bin/bash
# scatter tile prefetch: correctness subset, then A/B against a build without it (same box, interleaved)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_midscale.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_sharded.py -x -q > gpurun_out/r03_h_tests.log 2>&1; tail -3 gpurun_out/r03_h_tests.log
grep -q " failed" gpurun_out/r03_h_tests.log && exit 1
for wl in config3 uniform1b config2; do
  echo "#### $wl"
  scripts/ab_libs.sh "--no-extras --no-cpu-baseline --steps 5 --warmup 2 --workload $wl" librj_nopipe.so librj.so librj_nopipe.so librj.so
done > gpurun_out/r03_h_scatter_prefetch_ab.log 2>&1
cat gpurun_out/r03_h_scatter_prefetch_ab.log
