#!/bin/bash
# GPU box: tools/calib/fetch_calib under rocprofv3, one counter group per pass.  usage: tools/calib/run_calib.sh <tag>
[ -n "$1" ] || { echo "usage: $0 <tag>"; exit 2; }
R=${GRAFT_REPO_ROOT:-/root/repo}
O="$R/gpurun_out/$1"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
BIN="$R/jaderaytracerendering_amd/lib/fetch_calib"
timeout -k 10 120 "$BIN" > "$O/plain.jsonl" 2> "$O/plain.err" || { echo "fetch_calib failed"; tail -3 "$O/plain.err"; exit 1; }
cat "$O/plain.jsonl"
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*" | sort -u > "$O/counter_names.txt"
i=0
for grp in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$O/p$i" -- "$BIN" > "$O/p$i.log" 2>&1 || echo "pass $i ($grp) failed"
done
python3 - "$O" <<'PY'
import csv, glob, json, sys, collections
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
order = []
for f in sorted(glob.glob(O + "/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("k_") and not k.startswith("void k_"):
            continue
        # one table after the other: dispatch order tells which table a launch belongs to
        agg[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
json.dump({"%s#%s" % k: v for k, v in agg.items()}, open(O + "/counters_by_dispatch.json", "w"), indent=0)
print("dispatches with counters:", len(agg))
PY
rm -rf "$O"/p[0-9]
