#!/bin/bash
# round 3, GPU call 2: k_trace variants (prefetch, 5 waves/SIMD), fixed stall profile, concurrent shares, FETCH_SIZE calibration,
# the default bench line (parity_check, rooflines), Mray/s vs max_state_bytes, rank 0 of 8 on one GPU
R=${GRAFT_REPO_ROOT:-/root/repo}
O="$R/gpurun_out/r03_c2"
mkdir -p "$O"
cd "$R"
L="$R/jaderaytracerendering_amd/lib"
JADE_HIP_LIB=$L/libjade_hip_pref.so timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "trace_rays or small_configs or c1_cornell or missing_children or c5_deep" > "$O/pytest_pref.log" 2>&1
rc=$?; tail -3 "$O/pytest_pref.log"
if [ $rc -ge 124 ]; then echo "variant test timed out ($rc): stopping"; exit $rc; fi
timeout -k 10 400 python3 tools/ab_variants.py "" _pref _w5 _prefw5 > "$O/ab_c3.log" 2>&1; cat "$O/ab_c3.log" | tail -6
AB_CLOSEUP=1 timeout -k 10 400 python3 tools/ab_variants.py "" _pref _w5 _prefw5 > "$O/ab_closeup.log" 2>&1; tail -5 "$O/ab_closeup.log"
timeout -k 10 200 python3 tools/trace_profile.py --out "$O/trace_stalls_c3.json" > "$O/prof_c3.log" 2>&1 || { echo "profile failed"; tail -5 "$O/prof_c3.log"; }
timeout -k 10 300 python3 tools/concurrency_test.py --out "$O/concurrency.json" > "$O/concurrency.log" 2>&1; cat "$O/concurrency.log" | tail -5
bash tools/calib/run_calib.sh r03_c2/calib > "$O/calib.log" 2>&1; tail -12 "$O/calib.log"
timeout -k 10 400 python3 bench.py > "$O/bench.json" 2> "$O/bench.err" || { echo "bench failed"; tail -5 "$O/bench.err"; }
python3 - "$O/bench.json" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("bench:", round(d["value"]), "Mray/s", d["kernels"], "\nparity:", d.get("parity_check"), "\nstate:", d.get("state"), "\ncpu:", d.get("cpu_baseline"))
except Exception as e:
    print("bench: no line:", e)
PY
for gb in 32 64 128; do
  timeout -k 10 300 python3 bench.py --max-state-gb $gb --no-cpu-baseline --no-extras > "$O/bench_state_$gb.json" 2> "$O/bench_state_$gb.err" || echo "state $gb failed"
done
timeout -k 10 300 python3 bench.py --virtual-ranks 8 --no-cpu-baseline --no-extras > "$O/bench_virtual8.json" 2> "$O/bench_virtual8.err" || echo "virtual 8 failed"
python3 - "$O" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], round(d["value"]), "Mray/s", "ms/step %.1f" % d["ms_per_step"], d.get("state"))
    except Exception as e:
        print(f, "no line", e)
PY
