#!/bin/bash
set -e
mkdir -p gpurun_out/r03_w5
O=gpurun_out/r03_w5
timeout -k 10 900 python3 -m pytest tests/test_gpu_early_exit.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
export JADE_WIDE=1
AB_ROUNDS=3 timeout -k 10 400 python3 tools/ab_variants.py "" _nowide > $O/ab_c3.log 2>&1 && cat $O/ab_c3.log
AB_ROUNDS=3 AB_CLOSEUP=1 timeout -k 10 400 python3 tools/ab_variants.py "" _nowide > $O/ab_close.log 2>&1 && cat $O/ab_close.log
AB_CONFIG=C5 AB_ROUNDS=2 timeout -k 10 400 python3 tools/ab_variants.py "" _nowide > $O/ab_c5.log 2>&1 && cat $O/ab_c5.log
JADE_WIDE=1 timeout -k 10 300 python3 tools/trace_profile.py --out $O/stalls_c3_wide.json > $O/stalls_w.log 2>&1 || echo "wide profile failed"
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03_w5/stalls_c3_wide.json"))
print("wide Mray/s", round(d["Mray_per_s_profile_build"]), "units/ray", {k: round(v, 3) for k, v in d["wave_units_per_ray"].items()}, "walk", {k: round(v) for k, v in d["clocks_per_walk_unit"].items()})
PY
