#!/bin/bash
# carry-over threshold with early exits, whole-render view: 4 steps + flush
mkdir -p gpurun_out/r03_ee5
O=gpurun_out/r03_ee5
for f in 0.001 0.002 0.003 0.005 0.02; do
  JADE_CARRY_FRACTION=$f timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras > "$O/carry_$f.json" 2> "$O/carry_$f.err"
  python3 - "$O/carry_$f.json" "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); k = d["kernels"]
    print("carry %-6s %6.0f Mray/s  ms/step %6.1f  k_trace %6.1f  rest %6.1f  launches %d  flush %.0f ms  render(4 steps + flush) %.0f ms" % (sys.argv[2], d["value"], d["ms_per_step"], k["k_trace"]["ms_per_step"], k["rest_ms_per_step"], d["roofline"]["launches"], d["final_flush"]["ms"], 4 * d["ms_per_step"]))
except Exception as e:
    print("carry", sys.argv[2], "no result", e)
PY
done
timeout -k 10 300 python3 tools/trace_profile.py --out gpurun_out/r03_ee5/trace_stalls_c3.json > $O/stalls.log 2>&1 || echo "stall profile failed"
