#!/bin/bash
set -e
mkdir -p gpurun_out/r03_p1
O=gpurun_out/r03_p1
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
timeout -k 10 600 python3 bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err && python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03_p1/bench.json'))
print(round(d['value']), round(d['ms_per_step'],1), {k:(round(v['ms_per_step'],1) if isinstance(v,dict) else round(v,1)) for k,v in d['kernels'].items()}, d['statue_closeup']['value'])
PY
timeout -k 10 600 python3 bench.py --config C5 --steps 2 --warmup 1 --spp-per-step 64 --no-cpu-baseline --no-extras > $O/c5.json 2> $O/c5.err && python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03_p1/c5.json'))
print('C5', round(d['value']), round(d['ms_per_step'],1), {k:(round(v['ms_per_step'],1) if isinstance(v,dict) else round(v,1)) for k,v in d['kernels'].items()})
PY
