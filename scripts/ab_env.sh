#!/bin/bash
# A/B of runtime tuning knobs on one box: scripts/ab_env.sh "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...
args="$1"; shift
for e in "$@"; do
  echo "== $e"
  env $e timeout -k 10 200 python3 bench.py $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['roofline']['kernels']
print('%.2f G/s  %.2f ms/step  ' % (d['value']/1e9, d['ms_per_step']) + '  '.join('%s=%.2f' % (n, v['ms_per_step']) for n, v in k.items()))
" || exit 1
done
