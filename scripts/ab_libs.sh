#!/bin/bash
# A/B of tuning variants of librj.so on one box: scripts/ab_libs.sh "<bench args>" lib1.so lib2.so ...
# (variants are built with `make -C radix-join_amd/csrc LIB=../librj_x.so OBJDIR=build_x EXTRA=-D...`)
args="$1"; shift
for lib in "$@"; do
  echo "== $lib"
  RJ_LIB_PATH=$PWD/radix-join_amd/$lib timeout -k 10 200 python3 bench.py $args 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['roofline']['kernels']
print('%.2f G/s  %.2f ms/step  ' % (d['value']/1e9, d['ms_per_step']) + '  '.join('%s=%.2f' % (n, v['ms_per_step']) for n, v in k.items()))
" || exit 1
done
