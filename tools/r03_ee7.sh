#!/bin/bash
# environment-only knobs once more, with early exits: the packet budget, k_trace blocks per CU
mkdir -p gpurun_out/r03_ee7
O=gpurun_out/r03_ee7
run() {
  env "$@" timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras > "$O/$1.json" 2> "$O/$1.err"
  python3 - "$O/$1.json" "$*" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); k = d["kernels"]
    print("%-40s %6.0f Mray/s  ms/step %6.1f  k_trace %6.1f  k_light %6.1f  rest %6.1f" % (sys.argv[2], d["value"], d["ms_per_step"], k["k_trace"]["ms_per_step"], k["k_light"]["ms_per_step"], k["rest_ms_per_step"]))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
}
run JADE_PACKET_BUDGET=32
run JADE_PACKET_BUDGET=16
run JADE_PACKET_BUDGET=24
run JADE_PACKET_BUDGET=48
run JADE_PACKET_BUDGET=96
