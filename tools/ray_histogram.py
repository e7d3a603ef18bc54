#!/usr/bin/env python3
"""Development diagnostic: per-ray work and stack-depth histograms of a config, from the CPU oracle.
usage: ray_histogram.py [config] [spp] [width height]   (default C2 1 128 128)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jaderaytracerendering_amd as J  # noqa: E402
from jaderaytracerendering_amd import backend as B  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w = int(sys.argv[3]) if len(sys.argv) > 3 else 128
h = int(sys.argv[4]) if len(sys.argv) > 4 else 128
hs, cfg = J.build_config(name)
be = B.Backend(os.path.join(ROOT, "oracle", "libjade_oracle.so"))
sc = be.scene(hs)
visit = (ctypes.c_uint64 * 64)()
stack = (ctypes.c_uint64 * 64)()
be.lib.jade_oracle_visit_histogram(visit, 1)
be.lib.jade_oracle_stack_histogram(stack, 1)
sc.render(B.make_params(w, h, spp, list(cfg.eye), list(cfg.camera)))
be.lib.jade_oracle_visit_histogram(visit, 0)
be.lib.jade_oracle_stack_histogram(stack, 0)
v = np.array(visit[:], dtype=np.float64).reshape(2, 32)
s = np.array(stack[:], dtype=np.float64)
print("rays", int(s.sum()))
for k, nm in enumerate(("node pops", "tri tests+1")):
    print(nm, " ".join("2^%d:%.3f" % (b, v[k, b] / v[k].sum()) for b in range(32) if v[k, b]))
cum = np.cumsum(s) / s.sum()
print("deepest stack  " + " ".join("%d:%.4f" % (d, cum[d]) for d in range(64) if s[d]))
