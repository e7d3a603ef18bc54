#!/bin/bash
# the driver's round-end sequence, rehearsed: build check, GPU suite, smoke, default bench
set -e
mkdir -p gpurun_out/r03_verify
O=gpurun_out/r03_verify
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && cat $O/smoke.log
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 2 > $O/bench.json 2> $O/bench.err
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03_verify/bench.json') if l.startswith('{')][-1])
print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','scaling','dtype','binding','value_reference_walk')}, d['roofline']['frac'], d['cpu_baseline']['value'], d['parity_check']['ok'])
PY
