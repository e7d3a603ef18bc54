#!/bin/bash
# GPU box: k_trace time per step under different ray-queue orderings (host-followed passes: JADE_BATCH=0).
# usage: tools/sort_sweep.sh <outdir-under-gpurun_out>
[ -n "$1" ] || { echo "usage: $0 <tag>"; exit 2; }
R=${GRAFT_REPO_ROOT:-/root/repo}
O="$R/gpurun_out/$1"
mkdir -p "$O"
run() {  # name, env...
  local name=$1; shift
  env "$@" JADE_BATCH=0 JADE_LOG_SORT=1 timeout -k 10 300 python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$O/$name.json" 2> "$O/$name.err" || echo "$name failed"
  python3 - "$O/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d["kernels"]; c = d.get("statue_closeup") or {}
    print("%-22s %7.0f Mray/s  k_trace %6.1f ms/step (%5.0f Mray/s)  k_light %5.1f  device %6.1f ms/step | closeup %6.0f Mray/s (k_trace %5.0f)" % (
        sys.argv[2], d["value"], k["k_trace"]["ms_per_step"], k["k_trace"]["Mray_per_s"], k["k_light"]["ms_per_step"], k["device_ms_per_step"],
        c.get("value", 0), c.get("k_trace_Mray_per_s") or 0))
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
  grep "ray-queue" "$O/$name.err" | tail -2
}
run nosort JADE_SORT=0
run tri JADE_SORT=1
run oct_tri JADE_SORT=2
run tri_oct JADE_SORT=3
run tri8_oct JADE_SORT=3 JADE_SORT_TRISHIFT=3
run tri64_oct JADE_SORT=3 JADE_SORT_TRISHIFT=6
run cls_only JADE_SORT=1 JADE_SORT_BITS=3
run tri_top12 JADE_SORT=1 JADE_SORT_BITS=12
