#!/bin/bash
# GPU box: tools/calib/valu_calib plain and under rocprofv3 with the SQ issue counters DESIGN 3.4's "valu_busy" is made of.
# usage: tools/calib/run_valu_calib.sh <tag>   -> gpurun_out/<tag>/valu_calib.jsonl, valu_calib_counters.json
[ -n "$1" ] || { echo "usage: $0 <tag>"; exit 2; }
R=${GRAFT_REPO_ROOT:-/root/repo}
O="$R/gpurun_out/$1"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
BIN="$R/jaderaytracerendering_amd/lib/valu_calib"
timeout -k 10 120 "$BIN" > "$O/valu_calib.jsonl" 2> "$O/valu_calib.err" || { echo "valu_calib failed"; tail -3 "$O/valu_calib.err"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$O/pmc" -- "$BIN" > "$O/pmc.log" 2>&1 || { echo "pmc pass failed"; tail -5 "$O/pmc.log"; exit 1; }
python3 - "$O" <<'PY'
import csv, glob, json, sys, collections
O = sys.argv[1]
plain = [json.loads(l) for l in open(O + "/valu_calib.jsonl") if l.startswith("{")]
runs = [r for r in plain if "kernel" in r]
f = glob.glob(O + "/pmc/*/*counter_collection.csv")[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "k_valu" not in r["Kernel_Name"]:
        continue
    d = agg.setdefault(int(r["Dispatch_Id"]), {"kernel_name": r["Kernel_Name"], "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
timed = [agg[k] for k in sorted(agg)][1::2]  # every run is a warm-up launch followed by the timed one
out = []
for run, c in zip(runs, timed):
    busy = 4 * c["SQ_ACTIVE_INST_VALU"] / 1024 / (c["SQ_BUSY_CU_CYCLES"] / 256)
    out.append({"kernel": run["kernel"], "waves_per_simd": run["waves_per_simd"], "ms_plain": run["ms"], "ms_profiled": c["ns"] * 1e-6,
                "wave_insts_per_simd_per_ns": run["wave_insts_per_simd_per_ns"], "valu_busy": busy,
                "wave_insts_per_simd_per_busy_clock": c["SQ_INSTS_VALU"] / 1024 / (c["SQ_BUSY_CU_CYCLES"] / 256),
                "clock_GHz_under_load": (c["SQ_BUSY_CU_CYCLES"] / 256) / c["ns"]})
    print("%-45s w=%d  %.3f wave-insts/SIMD/ns  valu_busy %.3f  insts/SIMD/clock %.3f  clock %.2f GHz" % (
        run["kernel"], run["waves_per_simd"], run["wave_insts_per_simd_per_ns"], busy, out[-1]["wave_insts_per_simd_per_busy_clock"], out[-1]["clock_GHz_under_load"]))
json.dump({"command": "rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --kernel-trace -- valu_calib",
           "valu_busy": "4 x SQ_ACTIVE_INST_VALU / 1024 SIMDs / (SQ_BUSY_CU_CYCLES / 256 CUs), as in tools/summarize_prof.py", "runs": out},
          open(O + "/valu_calib_counters.json", "w"), indent=1)
PY
rm -rf "$O/pmc"
