#!/bin/bash
# round 3, GPU call 11: compact shading records (normal + material table) against the build before
R=${GRAFT_REPO_ROOT:-/root/repo}
O="$R/gpurun_out/r03_c11"; mkdir -p "$O"; cd "$R"
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "golden or parity or packet or edge or spec or scene_io or bench_schedule" > "$O/pytest.log" 2>&1
rc=$?; tail -3 "$O/pytest.log"
if [ $rc -ne 0 ]; then echo "parity failed: stopping"; tail -40 "$O/pytest.log"; exit $rc; fi
show() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d["kernels"]; c = d.get("statue_closeup") or {}
    print("%-14s %6.0f Mray/s  ms/step %6.1f  k_trace %6.1f  k_light %6.1f  rest %6.1f  rpp %d state %.0f GB | closeup %5.0f" % (
        sys.argv[2], d["value"], d["ms_per_step"], k["k_trace"]["ms_per_step"], k["k_light"]["ms_per_step"], k["rest_ms_per_step"], d["state"]["records_per_pixel"], d["state"]["state_bytes"] / 1e9, c.get("value", 0)))
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
A="--steps 3 --warmup 1 --no-cpu-baseline"
L="$R/jaderaytracerendering_amd/lib"
JADE_HIP_LIB=$L/libjade_hip_old.so timeout -k 10 300 python3 bench.py $A > "$O/old.json" 2> "$O/old.err"; show "$O/old.json" "before"
timeout -k 10 300 python3 bench.py $A > "$O/new.json" 2> "$O/new.err"; show "$O/new.json" "compact shading"
JADE_HIP_LIB=$L/libjade_hip_old.so timeout -k 10 300 python3 bench.py $A > "$O/old2.json" 2> "$O/old2.err"; show "$O/old2.json" "before (2)"
timeout -k 10 300 python3 bench.py $A > "$O/new2.json" 2> "$O/new2.err"; show "$O/new2.json" "compact shading (2)"
C5="--config C5 --steps 2 --warmup 1 --spp-per-step 64 --no-cpu-baseline --no-extras"
JADE_HIP_LIB=$L/libjade_hip_old.so timeout -k 10 300 python3 bench.py $C5 > "$O/c5_old.json" 2> "$O/c5_old.err"; show "$O/c5_old.json" "C5 before"
timeout -k 10 300 python3 bench.py $C5 > "$O/c5_new.json" 2> "$O/c5_new.err"; show "$O/c5_new.json" "C5 compact shading"
