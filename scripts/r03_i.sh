#!/bin/bash
# round-3 profiles: rocprofv3 kernel stats + FETCH/WRITE PMC passes for the three synthetic workloads, then the default bench
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for wl in config3 uniform1b config2; do
  scripts/profile_run.sh r03_zz_${wl} $wl || { echo "profile $wl failed"; tail -5 gpurun_out/prof_r03_zz_${wl}/*.log; exit 1; }
  echo "profiled $wl"
done
timeout -k 10 600 python bench.py > gpurun_out/r03_i_bench_default.json 2> gpurun_out/r03_i_bench_default.err || { tail -5 gpurun_out/r03_i_bench_default.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_i_bench_default.json").readlines()[-1])
print("config3 %.2f G/s %.2f ms" % (d["value"] / 1e9, d["ms_per_step"]), {k: round(v["ms_per_step"], 2) for k, v in d["roofline"]["kernels"].items()})
for k, v in d.get("configs", {}).items():
    print(k, "%.2f G/s %.2f ms" % (v["value"] / 1e9, v["ms_per_step"]))
print("plan_ms", {k: (round(v["ms"], 2), round(v["ms_first_call"], 2)) if isinstance(v, dict) else round(v, 2) for k, v in d.get("plan_ms", {}).items()})
print("cpu", d.get("cpu_baseline", {}).get("value"))
PY
