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
    f = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))[-1]
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
json.dump(summary, open(os.path.join(ROOT, "profiles", tag + "_pmc_summary.json"), "w"), indent=1)
json.dump({"k_trace_hbm_bytes_per_launch": 2 * fetch + write, "source": "profiles/%s_pmc_summary.json" % tag},
          open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"))
print({k: v for k, v in summary.items() if k.startswith("k_trace")})
