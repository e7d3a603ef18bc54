#!/bin/bash
# round 3, GPU call 9: ray ordering on C5 (opt-in), short-queue early exit, the suite's quick part
R=${GRAFT_REPO_ROOT:-/root/repo}
O="$R/gpurun_out/r03_c9"; mkdir -p "$O"; cd "$R"
timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "trace_rays or golden or small_configs or c1_cornell or carry or progressive" > "$O/pytest.log" 2>&1
rc=$?; tail -3 "$O/pytest.log"
if [ $rc -ne 0 ]; then echo "parity failed: stopping"; tail -30 "$O/pytest.log"; exit $rc; fi
JADE_SORT=1 JADE_SORT_MIN=64 timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "golden or small_configs or c1_cornell or c5_deep" > "$O/pytest_sort.log" 2>&1; echo "with JADE_SORT=1: rc=$?"; tail -2 "$O/pytest_sort.log"
show() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d["kernels"]
    print("%-26s %6.0f Mray/s  ms/step %7.1f  k_trace %7.1f (%5.0f)  k_light %6.1f  rest %6.1f  launches %d  flush syncs %d" % (
        sys.argv[2], d["value"], d["ms_per_step"], k["k_trace"]["ms_per_step"], k["k_trace"]["Mray_per_s"], k["k_light"]["ms_per_step"], k["rest_ms_per_step"], d["roofline"]["launches"], d["host_syncs_in_final_flush"]))
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
C5="--config C5 --steps 2 --warmup 1 --spp-per-step 64 --no-cpu-baseline --no-extras"
timeout -k 10 300 python3 bench.py $C5 > "$O/c5_batch.json" 2> "$O/c5_batch.err"; show "$O/c5_batch.json" "C5 default (batched)"
JADE_BATCH=0 timeout -k 10 300 python3 bench.py $C5 > "$O/c5_host.json" 2> "$O/c5_host.err"; show "$O/c5_host.json" "C5 host-followed"
JADE_SORT=1 timeout -k 10 300 python3 bench.py $C5 > "$O/c5_sort.json" 2> "$O/c5_sort.err"; show "$O/c5_sort.json" "C5 ordered queue"
JADE_SORT=1 JADE_SORT_MIN=1000000 timeout -k 10 300 python3 bench.py $C5 > "$O/c5_sort1m.json" 2> "$O/c5_sort1m.err"; show "$O/c5_sort1m.json" "C5 ordered (>= 1 M rays)"
C5B="--config C5 --steps 2 --warmup 1 --spp-per-step 256 --no-cpu-baseline --no-extras"
timeout -k 10 400 python3 bench.py $C5B > "$O/c5b_batch.json" 2> "$O/c5b_batch.err"; show "$O/c5b_batch.json" "C5 256spp default"
JADE_SORT=1 timeout -k 10 400 python3 bench.py $C5B > "$O/c5b_sort.json" 2> "$O/c5b_sort.err"; show "$O/c5b_sort.json" "C5 256spp ordered"
timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-extras > "$O/c3.json" 2> "$O/c3.err"; show "$O/c3.json" "C3 default 4x1024"
