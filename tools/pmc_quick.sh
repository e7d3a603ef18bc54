#!/bin/bash
# GPU box, development: one rocprofv3 --pmc pass over a short bench run, totals per kernel.  usage: tools/pmc_quick.sh "<counters>" [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmcq; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
[ -n "$1" ] || { echo "usage: $0 \"<counters>\" [bench args]"; exit 2; }
grp=$1; shift
timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p -- python3 $R/bench.py --steps 1 --warmup 1 --spp-per-step 256 --no-cpu-baseline --no-extras "$@" > $O/p.log 2>&1
python3 - <<PY
import csv,glob,collections
fs=glob.glob("$O/p/*/*counter_collection.csv")
agg=collections.defaultdict(float); ns=collections.defaultdict(int); seen=set()
for r in csv.DictReader(open(fs[0])):
    k=r["Kernel_Name"].split("(")[0]
    if not k.startswith("k_"): continue
    agg[(k,r["Counter_Name"])]+=float(r["Counter_Value"])
    if (k,r["Dispatch_Id"]) not in seen:
        seen.add((k,r["Dispatch_Id"])); ns[k]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for (k,c),v in sorted(agg.items()):
    if k in ("k_trace","k_light","k_shade"): print("%-9s %-34s %.5g   (kernel ns %.4g)"%(k,c,v,ns[k]))
PY
rm -rf $O/p
