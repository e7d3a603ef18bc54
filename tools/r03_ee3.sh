#!/bin/bash
# early exit: the whole GPU suite, then a bench line
set -e
mkdir -p gpurun_out/r03_ee3
O=gpurun_out/r03_ee3
timeout -k 10 1000 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && cat $O/smoke.log
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err && python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03_ee3/bench.json'))
print(d['value'], d['ms_per_step'], d['kernels'], d['reference_walk'], d['statue_closeup'], d['parity_check']['ok'], d['final_flush'], d['binding'])
PY
