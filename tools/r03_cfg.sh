#!/bin/bash
# the other configurations of BASELINE.json, for the record (they are parity-test cases, not the headline)
mkdir -p gpurun_out/r03_cfg
O=gpurun_out/r03_cfg
run() {
  name=$1; shift
  timeout -k 10 400 python3 bench.py "$@" > $O/$name.json 2> $O/$name.err
  python3 - $O/$name.json "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); k = d["kernels"]
    pc = d.get("parity_check") or {}
    print("%-22s %7.0f Mray/s  ms/step %7.2f  k_trace %7.2f  k_light %6.2f  rest %6.2f  ref-walk %s  parity %s  cpu %s" % (sys.argv[2], d["value"], d["ms_per_step"], k["k_trace"]["ms_per_step"], k["k_light"]["ms_per_step"], k["rest_ms_per_step"], d["reference_walk"] and round(d["reference_walk"]["value"]), pc.get("ok"), d.get("cpu_baseline") and round(d["cpu_baseline"]["value"], 2)))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
}
run C1_256x256x64 --config C1 --spp-per-step 64 --steps 4 --warmup 1
run C2_512x512x256 --config C2 --spp-per-step 256 --steps 4 --warmup 1
run C3_reference_walk --reference-walk --no-cpu-baseline --no-extras
