#!/bin/bash
# round 3, GPU call 5: the packet first pass that hands fanned-out packets to the wavefront passes: parity, budget and occupancy sweeps
R=${GRAFT_REPO_ROOT:-/root/repo}
O="$R/gpurun_out/r03_c5"
mkdir -p "$O"
cd "$R"
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "golden or parity or packet or edge or spec or scene_io or bench_schedule" > "$O/pytest.log" 2>&1
rc=$?; tail -4 "$O/pytest.log"
if [ $rc -ne 0 ]; then echo "pytest failed ($rc): skipping timings"; tail -40 "$O/pytest.log"; exit $rc; fi
JADE_PACKET_BUDGET=4 timeout -k 10 300 python3 -m pytest tests -m gpu -x -q -k "golden or small_configs or c1_cornell or nonsquare or schedule or carry" > "$O/pytest_b4.log" 2>&1; echo "budget 4 (most packets given up): rc=$?"; tail -2 "$O/pytest_b4.log"
show() { python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    k = d["kernels"]
    print("%-18s %6.0f Mray/s  ms/step %6.1f  k_trace %6.1f (%5.0f Mray/s)  k_light %6.1f (%5.0f Mray/s)  rest %6.1f" % (
        sys.argv[2], d["value"], d["ms_per_step"], k["k_trace"]["ms_per_step"], k["k_trace"]["Mray_per_s"], k["k_light"]["ms_per_step"], k["k_light"]["Mray_per_s"], k["rest_ms_per_step"]))
except Exception as e:
    print(sys.argv[2], "no result:", e)
PY
}
A="--steps 2 --warmup 1 --no-cpu-baseline --no-extras"
C5="--config C5 --steps 2 --warmup 1 --spp-per-step 64 --no-cpu-baseline --no-extras"
L="$R/jaderaytracerendering_amd/lib"
for b in 16 32 64 128; do
  JADE_PACKET_BUDGET=$b timeout -k 10 300 python3 bench.py $A > "$O/c3_b$b.json" 2> "$O/c3_b$b.err"; show "$O/c3_b$b.json" "C3 budget $b"
done
for v in _pk3 _pk5; do
  JADE_HIP_LIB=$L/libjade_hip$v.so timeout -k 10 300 python3 bench.py $A > "$O/c3$v.json" 2> "$O/c3$v.err"; show "$O/c3$v.json" "C3 $v (budget 48)"
done
JADE_LIGHT_PACKET=0 timeout -k 10 300 python3 bench.py $A > "$O/c3_fifo.json" 2> "$O/c3_fifo.err"; show "$O/c3_fifo.json" "C3 per-lane"
for b in 16 32 64; do
  JADE_PACKET_BUDGET=$b timeout -k 10 300 python3 bench.py $C5 > "$O/c5_b$b.json" 2> "$O/c5_b$b.err"; show "$O/c5_b$b.json" "C5 budget $b"
done
JADE_LIGHT_PACKET=0 timeout -k 10 300 python3 bench.py $C5 > "$O/c5_fifo.json" 2> "$O/c5_fifo.err"; show "$O/c5_fifo.json" "C5 per-lane"
timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > "$O/c3_closeup.json" 2> "$O/c3_closeup.err"
python3 - "$O/c3_closeup.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print("closeup (budget 48):", d.get("statue_closeup"))
PY
JADE_LOG_PASSES=1 timeout -k 10 300 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras 2>&1 | grep "first pass:" | tail -2
