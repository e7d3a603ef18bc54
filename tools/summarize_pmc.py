#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc CSVs of one build into profiles/<tag>_pmc_summary.json + profiles/hbm_traffic.json.

usage: summarize_pmc.py <tag> <fetch_dir> <write_dir> <l2_dir> <bench.json>
Units (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE
reads half of the bytes fetched, so it is doubled; one counter group per rocprofv3 pass."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, fetch_dir, write_dir, l2_dir, bench = sys.argv[1:6]
out = {}
for d in (fetch_dir, write_dir, l2_dir):
    f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)  # the newest run in that directory
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    for (kn, cn), v in agg.items():
        if kn in ("k_trace", "k_shade", "k_shade_lean"):
            out.setdefault(kn, {})[cn] = {"dispatches": v[0], "sum": v[1], "avg_per_launch": v[1] / v[0]}
b = json.load(open(bench))
t = out["k_trace"]


def run_line(d):
    """the bench line the profiled command itself printed (tools/profile_round.sh keeps it in <dir>.log)"""
    for line in reversed(open(d.rstrip("/") + ".log", errors="replace").read().splitlines()):
        if line.startswith("{") and '"roofline"' in line:
            return json.loads(line)
    return None


pr = run_line(fetch_dir)
fetch = t["FETCH_SIZE"]["avg_per_launch"] * 1024
write = t["WRITE_SIZE"]["avg_per_launch"] * 1024
summary = {
    "build": tag,
    "command": "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
               "--no-cpu-baseline   (one pass per group: FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum)",
    "units": "FETCH_SIZE/WRITE_SIZE in KiB; bytes = value*1024; gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM)",
    "counters": out,
    "k_trace_fetch_bytes_per_launch_raw": fetch,
    "k_trace_fetch_bytes_per_launch_corrected": 2 * fetch,
    "k_trace_write_bytes_per_launch": write,
    "k_trace_hbm_bytes_per_launch": 2 * fetch + write,
    "k_trace_l2_hit_rate": t["TCC_HIT_sum"]["sum"] / (t["TCC_HIT_sum"]["sum"] + t["TCC_MISS_sum"]["sum"]),
    "k_trace_algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"],
}
# The launches of the profiled command (1 step + flush) are not those of the default run (4 steps + flush: the
# tail passes come once per run), so the per-launch figure is carried over as HBM bytes per ALGORITHMIC byte of
# the same run; bench.py multiplies it with its own algorithmic bytes per launch.
ratio = None
if pr:
    alg_total = pr["roofline"]["algorithmic_bytes_per_launch"] * pr["roofline"]["launches"]
    hbm_total = (2 * t["FETCH_SIZE"]["sum"] + t["WRITE_SIZE"]["sum"]) * 1024
    ratio = hbm_total / alg_total
    summary["profiled_run"] = {"launches": pr["roofline"]["launches"], "algorithmic_bytes_total": alg_total,
                               "k_trace_hbm_bytes_total": hbm_total, "k_trace_dispatches_seen": t["FETCH_SIZE"]["dispatches"]}
    summary["k_trace_hbm_bytes_per_algorithmic_byte"] = ratio
json.dump(summary, open(os.path.join(ROOT, "profiles", tag + "_pmc_summary.json"), "w"), indent=1)
json.dump({"k_trace_hbm_bytes_per_algorithmic_byte": ratio, "k_trace_hbm_bytes_per_launch_of_profiled_run": 2 * fetch + write,
           "source": "profiles/%s_pmc_summary.json" % tag}, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"))
print({k: v for k, v in summary.items() if k.startswith("k_trace")})
