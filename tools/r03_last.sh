#!/bin/bash
mkdir -p gpurun_out/r03_last
O=gpurun_out/r03_last
timeout -k 10 600 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
timeout -k 10 600 python3 bench.py --config C5 --steps 2 --warmup 1 --spp-per-step 64 > $O/c5_bench.json 2> $O/c5_bench.err; tail -c 200 $O/c5_bench.json; echo
timeout -k 10 400 python3 bench.py --config C1 --spp-per-step 64 --steps 4 --warmup 1 --no-cpu-baseline > $O/c1.json 2> $O/c1.err
timeout -k 10 400 python3 bench.py --config C2 --spp-per-step 256 --steps 4 --warmup 1 --no-cpu-baseline > $O/c2.json 2> $O/c2.err
python3 - <<'PY'
import json
for f in ('bench','c5_bench','c1','c2'):
    d=json.loads([l for l in open('gpurun_out/r03_last/%s.json'%f) if l.startswith('{')][-1])
    print(f, round(d['value']), round(d['ms_per_step'],2), d['binding'], 'ref', d['reference_walk'] and (round(d['reference_walk']['value']), round(d['reference_walk']['ms_per_step'],2)), 'closeup', d.get('statue_closeup') and round(d['statue_closeup']['value']), 'parity', (d.get('parity_check') or {}).get('ok'))
PY
