#!/usr/bin/env python3
"""Development helper: per-kernel sums of the counters collected by tools/pmc_passes.sh.
usage: pmc_table.py <tag>"""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
tab = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_%s_g*" % tag, "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("k_"):
            continue
        tab[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
for k in sorted(tab):
    print(k)
    for c in sorted(tab[k]):
        print("  %-40s %18.0f  (%d dispatches, avg %.1f)" % (c, tab[k][c], n[k][c], tab[k][c] / n[k][c]))
