#!/bin/bash
mkdir -p gpurun_out/r03_w4
O=gpurun_out/r03_w4
JADE_WIDE=0 timeout -k 10 300 python3 tools/trace_profile.py --out $O/stalls_c3_binary.json > $O/stalls_b.log 2>&1 || echo "binary failed"
JADE_WIDE=1 timeout -k 10 300 python3 tools/trace_profile.py --out $O/stalls_c3_wide.json > $O/stalls_w.log 2>&1 || echo "wide failed"
python3 - <<'PY'
import json
for n in ("binary", "wide"):
    d = json.load(open("gpurun_out/r03_w4/stalls_c3_%s.json" % n))
    print(n, "Mray/s", round(d["Mray_per_s_profile_build"]), "units/ray", {k: round(v, 3) for k, v in d["wave_units_per_ray"].items()}, "lanes", round(d["lanes_per_walk_unit"], 1), round(d["lanes_per_test_unit"], 1))
    print("   walk", {k: round(v) for k, v in d["clocks_per_walk_unit"].items()}, "test", {k: round(v) for k, v in d["clocks_per_test_unit"].items()})
    print("   share", {k: round(v, 3) for k, v in d["share"].items()})
    print("   counts", d["counts"])
PY
