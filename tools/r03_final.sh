#!/bin/bash
# round 3: the final measurement set: whole GPU suite, then tools/profile_r03.sh (bench, kernel stats, PMC passes; C3 and C5)
R=${GRAFT_REPO_ROOT:-/root/repo}
O="$R/gpurun_out/r03_final_tests"; mkdir -p "$O"; cd "$R"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1
rc=$?; tail -4 "$O/pytest.log"
if [ $rc -ne 0 ]; then echo "suite failed ($rc)"; tail -40 "$O/pytest.log"; exit $rc; fi
bash tools/profile_r03.sh r03_final C5 > "$O/profile.log" 2>&1; tail -12 "$O/profile.log"
timeout -k 10 300 python3 tools/trace_profile.py --out "$R/gpurun_out/r03_final/trace_stalls_c3.json" > "$O/stalls.log" 2>&1 || echo "stall profile failed"
# carry-over threshold sweep (environment only)
for f in 0.005 0.01 0.05 0.1; do
  JADE_CARRY_FRACTION=$f timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras > "$O/carry_$f.json" 2> "$O/carry_$f.err"
  python3 - "$O/carry_$f.json" "$f" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); k = d["kernels"]
    print("carry %-6s %6.0f Mray/s  ms/step %6.1f  k_trace %6.1f  rest %6.1f  launches %d  flush %.0f ms" % (sys.argv[2], d["value"], d["ms_per_step"], k["k_trace"]["ms_per_step"], k["rest_ms_per_step"], d["roofline"]["launches"], d["final_flush"]["ms"]))
except Exception as e:
    print("carry", sys.argv[2], "no result", e)
PY
done
