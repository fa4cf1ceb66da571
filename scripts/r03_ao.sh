#!/bin/bash
# packed pairs between the passes in blocks of 16 keys + 16 carries (RJ_TUNE_BLOCKED_MID=1): the second histogram reads
# 4 instead of 8 bytes per tuple.  One verified run, then six processes per variant, 1 B uniform.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
echo "== verified run, RJ_TUNE_BLOCKED_MID=1"
RJ_TUNE_BLOCKED_MID=1 timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 --workload uniform1b 2>&1 | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%.2f G/s %.2f ms verified=%s' % (d['value']/1e9, d['ms_per_step'], d['config'].get('verified')))
    elif 'rror' in l or 'FAILED' in l: print(l.strip())
"
scripts/ab_reps.sh 6 "--no-extras --no-cpu-baseline --no-verify --steps 4 --warmup 1 --workload uniform1b" RJ_TUNE_BLOCKED_MID=0 RJ_TUNE_BLOCKED_MID=1 | grep -A4 "^####"
} > gpurun_out/r03_ao_blocked_mid_ab.log 2>&1
cat gpurun_out/r03_ao_blocked_mid_ab.log
