#!/bin/bash
# A/B tuning variants of librj in one GPU call: scripts/bench_variants.sh "v1 v2 v3" [steps]
# (variants are built with csrc/Makefile LIB=../librj_<v>.so EXTRA=...)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
STEPS=${2:-5}
for v in base $1; do
  # a variant is a library suffix (librj_<v>.so) or an env assignment list "A=1,B=2" run on librj.so
  lib=radix-join_amd/librj.so; envs=""
  if [[ "$v" == *=* ]]; then envs=$(echo "$v" | tr ',' ' '); elif [ "$v" != base ]; then lib=radix-join_amd/librj_$v.so; fi
  echo "== $v"
  env $envs RJ_LIB_PATH=$PWD/$lib timeout -k 10 300 python bench.py --steps $STEPS --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
r = d['roofline']
print('  %.2f G tuples/s  %.3f ms/step' % (d['value'] / 1e9, d['ms_per_step']))
print('  ' + '  '.join('%s=%.3f' % (k, v) for k, v in r['kernels_ms_per_step'].items() if v > 0.01))
" | tee -a gpurun_out/variants.log
done
