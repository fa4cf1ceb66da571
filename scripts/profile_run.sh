#!/bin/bash
# rocprofv3 kernel stats + the two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs, counters never
# together with a trace domain) of one bench.py workload, condensed by scripts/prof_summary.py.
# Run on the GPU box from the repo root:   scripts/profile_run.sh <tag> <workload>
# Leaves gpurun_out/<tag>.md, gpurun_out/<tag>_kernel_stats.csv and gpurun_out/traffic.json (the
# copy of profiles/traffic_r02.json with this workload's HBM bytes per launch refreshed).
set -e
tag=$1
wlname=$2
root=$PWD
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
args="--no-extras --no-cpu-baseline --no-verify --steps 3 --warmup 1 --workload $wlname"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" $args > "$out/trace.log" 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 "$root/bench.py" $args > "$out/fetch.log" 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 "$root/bench.py" $args > "$out/write.log" 2>&1
cd "$root"
[ -f gpurun_out/traffic.json ] || cp profiles/traffic_r03.json gpurun_out/traffic.json 2>/dev/null || cp profiles/traffic_r02.json gpurun_out/traffic.json
python3 scripts/prof_summary.py "$out" "gpurun_out/$tag.md" gpurun_out/traffic.json "$wlname" \
    "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py $args"
cp "$(find "$out/trace" -name '*kernel_stats.csv' | head -1)" "gpurun_out/${tag}_kernel_stats.csv"
# the raw per-dispatch CSVs are large: keep only the summaries
rm -rf "$out/trace" "$out/fetch" "$out/write"
