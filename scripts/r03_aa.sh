#!/bin/bash
# the second pass of the packed 1 B-row plan chunk by chunk through the Infinity Cache (RJ_TUNE_MALL_CHUNK segments per
# chunk, two streams) against one launch per kernel: first one verified run each, then six processes per variant
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in 0 8; do
  echo "== verified run, RJ_TUNE_MALL_CHUNK=$v"
  RJ_TUNE_MALL_CHUNK=$v timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 --workload uniform1b 2>&1 | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%.2f G/s %.2f ms verified=%s' % (d['value']/1e9, d['ms_per_step'], d['config'].get('verified')))
    elif 'rror' in l or 'FAILED' in l: print(l.strip())
" || exit 1
done > gpurun_out/r03_aa_mall_chunk_ab.log 2>&1
cat gpurun_out/r03_aa_mall_chunk_ab.log
scripts/ab_reps.sh 6 "--no-extras --no-cpu-baseline --no-verify --steps 4 --warmup 1 --workload uniform1b" RJ_TUNE_MALL_CHUNK=0 RJ_TUNE_MALL_CHUNK=4 RJ_TUNE_MALL_CHUNK=8 RJ_TUNE_MALL_CHUNK=16 RJ_TUNE_MALL_CHUNK=32 | grep -A8 "^####" >> gpurun_out/r03_aa_mall_chunk_ab.log 2>&1
tail -7 gpurun_out/r03_aa_mall_chunk_ab.log
