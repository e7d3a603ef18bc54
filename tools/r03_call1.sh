#!/bin/bash
# round 3, GPU call 1: the GPU test-suite, the k_trace stall profile, the ray-ordering sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
O="$R/gpurun_out/r03_c1"
mkdir -p "$O"
cd "$R"
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q -s > "$O/pytest.log" 2>&1
rc=$?
tail -5 "$O/pytest.log"
if [ $rc -ge 124 ]; then echo "pytest timed out / was killed ($rc): stopping"; exit $rc; fi
timeout -k 10 200 python3 tools/trace_profile.py --out "$O/trace_stalls_c3.json" > "$O/prof_c3.log" 2>&1 || { echo "profile failed"; tail -5 "$O/prof_c3.log"; }
timeout -k 10 200 python3 tools/trace_profile.py --closeup --out "$O/trace_stalls_closeup.json" > "$O/prof_closeup.log" 2>&1 || echo "closeup profile failed"
tools/sort_sweep.sh r03_c1/sort
exit $rc
