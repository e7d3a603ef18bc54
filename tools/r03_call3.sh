#!/bin/bash
# round 3, GPU call 3: the suite on the new default build (5 waves, packet first pass, BSSRDF guide table), then A/B of the first pass
R=${GRAFT_REPO_ROOT:-/root/repo}
O="$R/gpurun_out/r03_c3"
mkdir -p "$O"
cd "$R"
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1
rc=$?; tail -4 "$O/pytest.log"
if [ $rc -ge 124 ]; then echo "pytest timed out ($rc): stopping"; exit $rc; fi
if [ $rc -ne 0 ]; then echo "pytest failed: skipping timings"; tail -40 "$O/pytest.log"; exit $rc; fi
show() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d["kernels"]; c = d.get("statue_closeup") or {}
    print("%-14s %6.0f Mray/s  ms/step %6.1f  k_trace %6.1f  k_light %6.1f (%5.0f Mray/s, V/ray %.1f)  rest %6.1f | closeup %5.0f" % (
        sys.argv[2], d["value"], d["ms_per_step"], k["k_trace"]["ms_per_step"], k["k_light"]["ms_per_step"], k["k_light"]["Mray_per_s"], k["k_light"]["nodes_per_ray"], k["rest_ms_per_step"], c.get("value", 0)))
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
A="--steps 3 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 python3 bench.py $A > "$O/b_packet.json" 2> "$O/b_packet.err"; show "$O/b_packet.json" packet
JADE_LIGHT_PACKET=0 timeout -k 10 300 python3 bench.py $A > "$O/b_fifo.json" 2> "$O/b_fifo.err"; show "$O/b_fifo.json" per-lane
JADE_HIP_LIB=$R/jaderaytracerendering_amd/lib/libjade_hip_pk4.so timeout -k 10 300 python3 bench.py $A > "$O/b_pk4.json" 2> "$O/b_pk4.err"; show "$O/b_pk4.json" packet-4waves
C5="--config C5 --steps 2 --warmup 1 --spp-per-step 64 --no-cpu-baseline --no-extras"
timeout -k 10 300 python3 bench.py $C5 > "$O/c5_packet.json" 2> "$O/c5_packet.err"; show "$O/c5_packet.json" C5-packet
JADE_LIGHT_PACKET=0 timeout -k 10 300 python3 bench.py $C5 > "$O/c5_fifo.json" 2> "$O/c5_fifo.err"; show "$O/c5_fifo.json" C5-per-lane
