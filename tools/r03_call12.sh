#!/bin/bash
# round 3, GPU call 12: the triangle side of "staged through LDS", measured: a test unit's pair records fetched cooperatively into an LDS stage
R=${GRAFT_REPO_ROOT:-/root/repo}
O="$R/gpurun_out/r03_c12"; mkdir -p "$O"; cd "$R"
JADE_HIP_LIB=$R/jaderaytracerendering_amd/lib/libjade_hip_coop.so timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "trace_rays or golden or small_configs or c1_cornell or c5_deep" > "$O/pytest_coop.log" 2>&1
rc=$?; tail -2 "$O/pytest_coop.log"
if [ $rc -ne 0 ]; then echo "variant parity failed: stopping"; tail -30 "$O/pytest_coop.log"; exit $rc; fi
AB_ROUNDS=3 timeout -k 10 500 python3 tools/ab_variants.py "" _coop > "$O/ab_c3.log" 2>&1; tail -3 "$O/ab_c3.log"
JADE_TRACE_BLOCKS_PER_CU=3 AB_ROUNDS=3 timeout -k 10 500 python3 tools/ab_variants.py "" _coop > "$O/ab_c3_3blocks.log" 2>&1; echo "both at 3 blocks per CU:"; tail -2 "$O/ab_c3_3blocks.log"
AB_CLOSEUP=1 AB_ROUNDS=2 timeout -k 10 500 python3 tools/ab_variants.py "" _coop > "$O/ab_closeup.log" 2>&1; tail -2 "$O/ab_closeup.log"
C5="--config C5 --steps 2 --warmup 1 --spp-per-step 64 --no-cpu-baseline --no-extras"
for v in "" _coop; do
  JADE_HIP_LIB=$R/jaderaytracerendering_amd/lib/libjade_hip$v.so timeout -k 10 300 python3 bench.py $C5 > "$O/c5$v.json" 2> "$O/c5$v.err"
  python3 - "$O/c5$v.json" "C5 ${v:-base}" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); k = d["kernels"]
    print("%-10s %6.0f Mray/s  k_trace %6.1f ms (%5.0f Mray/s)" % (sys.argv[2], d["value"], k["k_trace"]["ms_per_step"], k["k_trace"]["Mray_per_s"]))
except Exception as e:
    print(sys.argv[2], "no result", e)
PY
done
