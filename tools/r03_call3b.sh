#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O="$R/gpurun_out/r03_c3b"; mkdir -p "$O"; cd "$R"
JADE_LIGHT_PACKET=0 timeout -k 10 300 python3 -m pytest tests/test_golden.py tests/test_gpu_parity.py -m gpu -q -k "golden or small_configs or c1_cornell" > "$O/fifo.log" 2>&1; echo "per-lane first pass: rc=$?"; tail -3 "$O/fifo.log"
timeout -k 10 300 python3 -m pytest tests/test_golden.py tests/test_gpu_parity.py -m gpu -q -k "golden or small_configs or c1_cornell" > "$O/packet.log" 2>&1; echo "packet first pass: rc=$?"; tail -3 "$O/packet.log"
grep -E "^E  " "$O/packet.log" | head -20
python3 - <<'PY'
import numpy as np, sys
sys.path.insert(0, ".")
import jaderaytracerendering_amd as J
from jaderaytracerendering_amd import backend as B
import os
oracle = B.Backend("oracle/libjade_oracle.so")
hip = J.hip()
for name, size, spp in (("tiny", 64, 1), ("tinyjade", 64, 1)):
    hs, cfg = J.build_config(name)
    p = B.params_from_config(cfg, spp=spp); p.width = p.height = size
    with oracle.scene(hs) as so, hip.scene(hs) as sh:
        ro, bo, so_ = so.render(p); rh, bh, sh_ = sh.render(p)
    print(name, "oracle", so_.rays_primary, so_.rays_secondary, so_.nodes_visited, so_.tris_tested, so_.shaded_hits, "| hip", sh_.rays_primary, sh_.rays_secondary, sh_.nodes_visited, sh_.tris_tested, sh_.shaded_hits, "inline", sh_.rays_inline, sh_.nodes_inline, sh_.tris_inline)
    d = np.abs(ro - rh).max(axis=2); print("  pixels differing:", int((d > 1e-6 * (np.abs(ro).max(axis=2) + 1e-9)).sum()), "of", d.size)
PY
