#!/bin/bash
# interleaved repetitions of bench.py variants with means: every process draws fresh scratch allocations, whose
# placement moves a step by +-0.5 ms, so single runs do not separate variants that differ by less
#   scripts/ab_reps.sh <reps> "<bench args>" "ENV=a" "ENV=b" ...   (an env string may be RJ_X=label)
reps=$1; shift
args="$1"; shift
tmp=$(mktemp)
for i in $(seq 1 "$reps"); do scripts/ab_env.sh "$args" "$@" || exit 1; done > "$tmp" 2>&1
cat "$tmp"
python3 - "$tmp" <<'PY'
import re, sys, collections, statistics as st
acc = collections.OrderedDict(); cur = None
for line in open(sys.argv[1]):
    if line.startswith("=="):
        cur = line[3:].strip()
        continue
    m = re.match(r"([\d.]+) G/s\s+([\d.]+) ms/step\s+(.*)", line)
    if m:
        k = {a: float(b) for a, b in (x.split("=") for x in m.group(3).split())}
        k["step"] = float(m.group(2))
        acc.setdefault(cur, []).append(k)
print("#### means over the repetitions (ms per step)")
for name, rows in acc.items():
    cols = list(rows[0])
    print("%-70s n=%d  " % (name.split("/")[-1], len(rows)) + "  ".join("%s=%.2f" % (c, st.mean(r[c] for r in rows)) for c in ["step"] + [c for c in cols if c != "step"]) + "  step_sd=%.2f" % st.pstdev(r["step"] for r in rows))
PY
rm -f "$tmp"
