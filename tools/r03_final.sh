#!/bin/bash
# round 3: the final measurement set: whole GPU suite, then tools/profile_r03.sh (bench, kernel stats, PMC passes; C3 and C5)
R=${GRAFT_REPO_ROOT:-/root/repo}
O="$R/gpurun_out/r03_final_tests"; mkdir -p "$O"; cd "$R"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1
rc=$?; tail -4 "$O/pytest.log"
if [ $rc -ne 0 ]; then echo "suite failed ($rc)"; tail -40 "$O/pytest.log"; exit $rc; fi
bash tools/profile_r03.sh r03_final C5 > "$O/profile.log" 2>&1; tail -12 "$O/profile.log"
timeout -k 10 300 python3 tools/trace_profile.py --out "$R/gpurun_out/r03_final/trace_stalls_c3.json" > "$O/stalls.log" 2>&1 || echo "stall profile failed"
