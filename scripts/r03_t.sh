#!/bin/bash
# ten interleaved repetitions: shipped build 9+9, 1024-digit build 9+9, 1024-digit build 8+10
# (every bench process draws fresh scratch allocations, whose placement moves the scatters by up to a millisecond)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
C10="RJ_LIB_PATH=$PWD/radix-join_amd/librj_cap10.so"
for W in config3 uniform1b; do
  A="--no-extras --no-cpu-baseline --no-verify --steps 4 --warmup 1 --workload $W"
  echo "#### $W"
  for i in 1 2 3 4 5 6 7 8 9 10; do
    scripts/ab_env.sh "$A" RJ_X=shipped "$C10" "$C10 RJ_TUNE_P1_BITS=8" || exit 1
  done
done > gpurun_out/r03_t_bits_8_10_reps.log 2>&1
python3 - <<'PY'
import re, collections
cur=None; w=None; acc=collections.defaultdict(list)
for line in open("gpurun_out/r03_t_bits_8_10_reps.log"):
    if line.startswith("####"): w=line.split()[1]
    elif line.startswith("=="): cur=line[3:].strip().replace("RJ_LIB_PATH=","").split("/")[-1]
    else:
        m=re.match(r"([\d.]+) G/s\s+([\d.]+) ms/step\s+(.*)", line)
        if m:
            k=dict(x.split("=") for x in m.group(3).split())
            acc[(w,cur)].append((float(m.group(2)), float(k["pass1_scatter"]), float(k["pass2_scatter"])))
for key,v in acc.items():
    n=len(v); import statistics as st
    print(key, "n=%d step %.2f±%.2f  p1s %.2f  p2s %.2f  (min step %.2f)" % (n, st.mean(x[0] for x in v), st.pstdev(x[0] for x in v), st.mean(x[1] for x in v), st.mean(x[2] for x in v), min(x[0] for x in v)))
PY
