#!/bin/bash
# GPU box: every rank's share of an 8-rank frame (tiles dealt (tx + ty) % 8, 8 x 1024 spp per step), one after the other on the one GPU
# (bench.py --virtual-ranks 8 --virtual-rank r): what the slowest of eight GPUs would take per step.  -> gpurun_out/r04_shares/
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04_shares; mkdir -p $O
for r in 0 1 2 3 4 5 6 7; do
  timeout -k 10 200 python3 bench.py --virtual-ranks 8 --virtual-rank $r --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/rank$r.json 2> $O/rank$r.err || { tail -5 $O/rank$r.err; exit 1; }
  python3 - $O/rank$r.json $r <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); k=d["kernels"]
print("rank %s of 8  %6.0f Mray/s  ms/step %6.1f  k_trace %6.1f (%5.0f Mray/s)  first pass %6.1f  rest %6.1f  rays %.4g" % (sys.argv[2], d["value"], d["ms_per_step"], k["k_trace"]["ms_per_step"], k["k_trace"]["Mray_per_s"], k["k_light"]["ms_per_step"], k["rest_ms_per_step"], d["rays"]))
PY
done
