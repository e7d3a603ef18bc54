#!/usr/bin/env python3
"""Development aid of the test suite (not collected): raw hitBVH queries, HIP module vs the CPU oracle, first mismatches printed.
usage: python tests/dbg_trace.py [config] [rays]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jaderaytracerendering_amd as J
from jaderaytracerendering_amd import backend as B
name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
hs, cfg = J.build_config(name)
rng = np.random.default_rng(1234)
v = hs.vertices()
lo, hi = v.reshape(-1, 3).min(0), v.reshape(-1, 3).max(0)
o = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
skip = np.full(n, -1, np.int32)
ora = B.Backend(os.path.join(ROOT, "oracle", "libjade_oracle.so"))
hip = B.hip()
with ora.scene(hs) as so, hip.scene(hs) as sh:
    i_o, t_o, p_o, st_o = so.trace_rays(o, d, skip)
    for rep in range(2):
        i_h, t_h, p_h, st_h = sh.trace_rays(o, d, skip)
        bad = np.nonzero(i_o != i_h)[0]
        print("rep", rep, "rays", n, "index mismatches", len(bad), "dist mismatches", int((t_o.view(np.uint32) != t_h.view(np.uint32)).sum()),
              "V", st_o.nodes_visited, st_h.nodes_visited, "T", st_o.tris_tested, st_h.tris_tested, flush=True)
        for j in bad[:10]:
            print("  ray", j, "oracle", i_o[j], t_o[j], "hip", i_h[j], t_h[j])
