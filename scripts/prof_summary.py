#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) to the rj:: kernels.

usage: prof_summary.py <gpurun_out/prof dir> <out.md> [<traffic.json> <workload> "<command profiled>"]
  with the last three arguments the HBM bytes per launch of the hot kernels (2 x FETCH_SIZE +
  WRITE_SIZE, gfx950 correction of MI355X_MICROARCH.md §HBM) are merged into <traffic.json>
  under key <workload> — the file bench.py replays as `roofline.traffic`
  expects  <dir>/trace/**/_kernel_stats.csv   (--kernel-trace --stats)   or **/*_results.db (rocpd)
           <dir>/fetch/**/_counter_collection.csv (--pmc FETCH_SIZE)   [optional]
           <dir>/write/**/_counter_collection.csv (--pmc WRITE_SIZE)   [optional]
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"rj::(k_[a-z_0-9]+)(<[^>]*(?:<[^>]*>)?[^>]*>)?", name)
    if not m:
        return None
    return m.group(1) + (m.group(2) or "")


def bench_name(sym):
    """rocprof kernel symbol -> the name bench.py / the library's profiler uses"""
    if sym.startswith("k_join<"):
        return "join_build_probe"
    if sym.startswith("k_fine_hist") or (sym.startswith("k_pass_hist") and "SrcLoader" in sym):
        return "pass1_hist"
    if sym.startswith("k_pass_hist"):
        return "pass2_hist"
    if sym.startswith("k_pass_scatter") and "SrcLoader" in sym:
        return "pass1_scatter"
    if sym.startswith("k_pass_scatter"):
        return "pass2_scatter"
    return None


def _dbs(d, sub):
    return glob.glob(os.path.join(d, sub, "**", "*_results.db"), recursive=True)


def kernel_stats(d):
    rows = []
    # rocprofv3's default output is a rocpd SQLite database; CSV only with --output-format csv
    for f in _dbs(d, "trace"):
        import sqlite3

        acc = defaultdict(list)
        for name, dur in sqlite3.connect(f).execute("select name, duration from kernels"):
            s = short(name)
            if s:
                acc[s].append(float(dur))
        for s, v in acc.items():
            rows.append((s, len(v), sum(v), sum(v) / len(v), min(v), max(v)))
    for f in glob.glob(os.path.join(d, "trace", "**", "*_kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            s = short(r["Name"])
            if s:
                rows.append((s, int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]), float(r["MinNs"]), float(r["MaxNs"])))
    return sorted(rows, key=lambda x: -x[2])


def pmc(d, sub, counter):
    acc = defaultdict(list)
    for f in _dbs(d, sub):
        import sqlite3

        q = "select kernel_name, value from counters_collection where counter_name = ?"
        for name, val in sqlite3.connect(f).execute(q, (counter,)):
            s = short(name)
            if s:
                acc[s].append(float(val))
    for f in glob.glob(os.path.join(d, sub, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            s = short(r["Kernel_Name"])
            if s and r["Counter_Name"] == counter:
                acc[s].append(float(r["Counter_Value"]))
    return acc


def main():
    d, out = sys.argv[1], sys.argv[2]
    ks = kernel_stats(d)
    fetch = pmc(d, "fetch", "FETCH_SIZE")
    write = pmc(d, "write", "WRITE_SIZE")
    lines = ["# rocprofv3 summary (rj:: kernels only)", "",
             "| kernel | calls | total ms | avg us | min us | max us |", "|---|---|---|---|---|---|"]
    for s, calls, tot, avg, mn, mx in ks:
        lines.append(f"| `{s}` | {calls} | {tot/1e6:.3f} | {avg/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} |")
    if fetch or write:
        lines += ["", "PMC (separate passes; FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; the top-N values are the",
                  "full-size launches).  On gfx950 FETCH_SIZE reports HALF of a wide coalesced read",
                  "(MI355X_MICROARCH.md §HBM) — `2x` column; narrower accesses are uncalibrated.", "",
                  "| kernel | dispatches | max FETCH_SIZE MiB | x2 MiB | max WRITE_SIZE MiB |", "|---|---|---|---|---|"]
        for s in sorted(set(fetch) | set(write)):
            fv = max(fetch.get(s, [0])) / 1024.0
            wv = max(write.get(s, [0])) / 1024.0
            lines.append(f"| `{s}` | {len(fetch.get(s, write.get(s, [])))} | {fv:.1f} | {2*fv:.1f} | {wv:.1f} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    if len(sys.argv) >= 6:
        import json

        tpath, workload, cmd = sys.argv[3], sys.argv[4], sys.argv[5]
        per = {}
        for s in sorted(set(fetch) | set(write)):
            name = bench_name(s)
            if not name:
                continue
            fv, wv = max(fetch.get(s, [0])), max(write.get(s, [0]))
            per[name] = {"bytes": (2 * fv + wv) * 1024.0, "read_bytes": 2 * fv * 1024.0, "write_bytes": wv * 1024.0, "kernel": s}
        try:
            allw = json.load(open(tpath))
        except Exception:
            allw = {}
        allw[workload] = {
            "source": f"profiles/{os.path.basename(tpath)} (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of `{cmd}`; "
                      "HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes: gfx950 reports half of a wide coalesced read, MI355X_MICROARCH.md §HBM)",
            "bytes_per_launch": {k: v["bytes"] for k, v in per.items()},
            "detail": per,
        }
        json.dump(allw, open(tpath, "w"), indent=1)
        print("traffic ->", tpath, workload)


if __name__ == "__main__":
    main()


def pmc_table(d, sub, counters):
    """mean per-dispatch value of each counter for the full-size launches of every rj:: kernel"""
    import statistics
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, sub, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            s = short(r["Kernel_Name"])
            if s:
                acc[s][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = []
    for k, cs in acc.items():
        row = [k]
        for c in counters:
            v = cs.get(c, [])
            # the largest values belong to the full-size launches
            v = sorted(v)[-max(1, len(v) // 2):] if v else [0]
            row.append(statistics.mean(v))
        rows.append(row)
    return rows
