#!/usr/bin/env python3
"""CPU experiment (oracle diagnostics only): what does a walk that prunes children by the nearest hit so far visit, and does
the frame keep its bits?  usage: prune_probe.py [C3|C5|C1|C2] [width height spp]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import jaderaytracerendering_amd as J  # noqa: E402
from jaderaytracerendering_amd import _abi, backend as B  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
W, H, SPP = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (480, 270, 8)
hs, cfg = J.build_config(name)
import subprocess  # noqa: E402
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "probe"])
be = B.Backend(os.path.join(ROOT, "oracle", "libjade_oracle_probe.so"))  # the probe build: libjade_oracle.so has no such walk
lib = be.lib
lib.jade_oracle_set_prune.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float]
lib.jade_oracle_set_prune.restype = None
lib.jade_oracle_prune_counters.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
lib.jade_oracle_prune_counters.restype = None
eye = list(cfg.eye)
if os.environ.get("AB_CLOSEUP"):
    centre = hs.vertices()[hs.tri_i32()[:, 0] == 0].reshape(-1, 3).mean(0)
    eye = [float(x) for x in centre - 0.22 * (-np.array(cfg.camera[8:11], np.float32))]
p = B.make_params(W, H, SPP, eye, list(cfg.camera))
with be.scene(hs) as sc:
    t = time.time()
    rgb0, bgr0, st0 = sc.render(p)
    print("exhaustive: rays %d  V/ray %.1f (internal pops/ray %.1f)  T/ray %.1f  %.1f s" % (st0.rays, st0.nodes_visited / st0.rays,
          (st0.nodes_visited / st0.rays - 1) / 2, st0.tris_tested / st0.rays, time.time() - t), flush=True)
    for mode, rel, ab, minz in [(2, 1e-4, 1e-5, 0.0), (1, 1e-4, 1e-5, 0.0), (3, 0, 0, 0), (4, 1e-4, 1e-5, 0.0)]:
        lib.jade_oracle_set_prune(mode, rel, ab, minz)
        buf = (ctypes.c_uint64 * 4)()
        lib.jade_oracle_prune_counters(buf, 1)
        rgb, bgr, st = sc.render(p)
        lib.jade_oracle_prune_counters(buf, 1)
        lib.jade_oracle_set_prune(0, 0, 0, 0)
        same = np.array_equal(rgb.view(np.uint32), rgb0.view(np.uint32))
        ndiff = int((rgb.view(np.uint32) != rgb0.view(np.uint32)).any(axis=-1).sum())
        print("mode %d rel %g abs %g min|dz| %g: rays %d (%s)  internal pops/ray %.1f  leaves/ray %.1f  T/ray %.1f  frame %s (%d pixels differ)" % (
            mode, rel, ab, minz, st.rays, "equal" if st.rays == st0.rays else "DIFFERENT", buf[0] / buf[3], buf[1] / buf[3], buf[2] / buf[3],
            "bit-identical" if same else "DIFFERS", ndiff), flush=True)
