#!/bin/bash
# round 3, GPU call 7: the whole GPU suite on the new default build, then the bench at two step sizes
R=${GRAFT_REPO_ROOT:-/root/repo}
O="$R/gpurun_out/r03_c7"
mkdir -p "$O"
cd "$R"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --durations=8 > "$O/pytest.log" 2>&1
rc=$?; tail -14 "$O/pytest.log"
if [ $rc -ge 124 ]; then echo "pytest timed out ($rc): stopping"; exit $rc; fi
show() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d["kernels"]; c = d.get("statue_closeup") or {}
    print("%-22s %6.0f Mray/s  ms/step %7.1f  k_trace %7.1f (%5.0f)  k_light %6.1f (%5.0f)  rest %6.1f  launches %d syncs/step %.1f flush syncs %d | closeup %5.0f | parity %s" % (
        sys.argv[2], d["value"], d["ms_per_step"], k["k_trace"]["ms_per_step"], k["k_trace"]["Mray_per_s"], k["k_light"]["ms_per_step"], k["k_light"]["Mray_per_s"], k["rest_ms_per_step"],
        d["roofline"]["launches"], d["host_syncs_per_step"], d["host_syncs_in_final_flush"], c.get("value", 0), (d.get("parity_check") or {}).get("ok")))
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
timeout -k 10 400 python3 bench.py > "$O/bench.json" 2> "$O/bench.err"; show "$O/bench.json" "default 4x1024"
timeout -k 10 400 python3 bench.py --spp-per-step 4096 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$O/bench_4096.json" 2> "$O/bench_4096.err"; show "$O/bench_4096.json" "2x4096"
timeout -k 10 400 python3 bench.py --spp-per-step 2048 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$O/bench_2048.json" 2> "$O/bench_2048.err"; show "$O/bench_2048.json" "2x2048"
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 --no-extras > "$O/bench_20.json" 2> "$O/bench_20.err"; show "$O/bench_20.json" "driver form 20x1024"
