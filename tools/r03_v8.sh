#!/bin/bash
mkdir -p gpurun_out/r03_v8
O=gpurun_out/r03_v8
for g in 0 45 25; do
  timeout -k 10 300 python3 bench.py --virtual-ranks 8 --no-cpu-baseline --no-extras --max-state-gb $g > $O/v8_$g.json 2> $O/v8_$g.err
  python3 - $O/v8_$g.json $g <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); k = d["kernels"]
print("max-state-gb %-3s rpp %4d  %6.0f Mray/s  ms/step %6.1f  k_trace %6.1f (%5.0f Mray/s, %d launches)  k_light %6.1f  rest %6.1f  rays %.3g" % (sys.argv[2], d["state"]["records_per_pixel"], d["value"], d["ms_per_step"], k["k_trace"]["ms_per_step"], k["k_trace"]["Mray_per_s"], d["roofline"]["launches"], k["k_light"]["ms_per_step"], k["rest_ms_per_step"], d["rays"]))
PY
done
for r in 1 2 3 4 5 6 7; do
  timeout -k 10 300 python3 bench.py --virtual-ranks 8 --virtual-rank $r --no-cpu-baseline --no-extras > $O/v8_rank$r.json 2> $O/v8_rank$r.err
  python3 - $O/v8_rank$r.json $r <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); k = d["kernels"]
print("rank %s of 8  %6.0f Mray/s  ms/step %6.1f  k_trace %6.1f (%5.0f Mray/s)  k_light %6.1f  rest %6.1f  rays %.4g" % (sys.argv[2], d["value"], d["ms_per_step"], k["k_trace"]["ms_per_step"], k["k_trace"]["Mray_per_s"], k["k_light"]["ms_per_step"], k["rest_ms_per_step"], d["rays"]))
PY
done
