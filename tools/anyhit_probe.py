#!/usr/bin/env python3
"""CPU experiment (oracle probe build only; diagnostics, never loaded by tests / smoke / bench): what the early-exit queries cost
by kind and outcome, what a last-occluder cache per (source triangle, query) would answer without a walk, and what a walk that
takes the larger child first visits.  usage: anyhit_probe.py [C3|C5|C2] [width height spp]   (AB_CLOSEUP=1: the statue close-up)"""
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import jaderaytracerendering_amd as J  # noqa: E402
from jaderaytracerendering_amd import backend as B  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
W, H, SPP = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (480, 270, 8)
hs, cfg = J.build_config(name)
subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "probe"])
be = B.Backend(os.path.join(ROOT, "oracle", "libjade_oracle_probe.so"))
lib = be.lib
lib.jade_oracle_set_prune.argtypes = [ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float]
lib.jade_oracle_set_prune.restype = None
lib.jade_oracle_kind_counters.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
lib.jade_oracle_kind_counters.restype = None
eye = list(cfg.eye)
if os.environ.get("AB_CLOSEUP"):
    centre = hs.vertices()[hs.tri_i32()[:, 0] == 0].reshape(-1, 3).mean(0)
    eye = [float(x) for x in centre - 0.22 * (-np.array(cfg.camera[8:11], np.float32))]
p = B.make_params(W, H, SPP, eye, list(cfg.camera))
KIND = ["camera/mirror", "shadow", "environment", "indirect"]
with be.scene(hs) as sc:
    rgb0, bgr0, st0 = sc.render(p)
    print("%s %dx%dx%d reference walk: rays %d  V/ray %.1f  T/ray %.1f" % (name, W, H, SPP, st0.rays, st0.nodes_visited / st0.rays, st0.tris_tested / st0.rays))
    lib.jade_oracle_occ_reset.argtypes = []
    lib.jade_oracle_occ_reset.restype = None
    variants = [(-1, 0, 1, "early exits"), (0, 0, 1, "triangle"), (0, 1, 1, "triangle, larger first"), (0, 1, 4, "4 triangles, larger first"),
                (1, 0, 1, "subtree 1 up"), (2, 0, 1, "subtree 2 up"), (2, 1, 1, "subtree 2 up, larger first"), (3, 0, 1, "subtree 3 up"), (2, 0, 2, "2 subtrees 2 up")]
    if os.environ.get("SWEEP"):
        variants = [(-1, 0, 1, "early exits")] + [(u, 0, w, "subtree %d up, %d ways" % (u, w)) for u, w in [(1, 1), (1, 2), (1, 4), (2, 1), (2, 2), (2, 4), (3, 2), (3, 4), (4, 2), (4, 4)]]
    if os.environ.get("PICK"):
        variants = [(-1, 0, 1, "early exits"), (1, 0, 4, "subtree 1 up, 4 ways"), (2, 0, 4, "subtree 2 up, 4 ways"), (1, 1, 4, "subtree 1 up, 4 ways, larger child first")]
    if os.environ.get("KEYS"):
        variants = [(1, 0, 4, "1 up, 4 ways, env by octant"), (1, 4, 4, "1 up, 4 ways, env one key"), (1, 8, 4, "1 up, 4 ways, env by sign of d.y")]
    if os.environ.get("VARIANTS"):
        variants = [variants[int(i)] for i in os.environ["VARIANTS"].split(",")]
    warm = int(os.environ.get("WARM", "0"))
    for up, flags, ways, what in variants:
        buf = (ctypes.c_uint64 * 40)()
        lib.jade_oracle_occ_reset()
        lib.jade_oracle_set_prune(5, up, flags, ways)
        for i in range(warm):  # other samples of the same frame (another RNG frame counter): a replay of the SAME rays would find each one's own answer cached
            pw = type(p).from_buffer_copy(p)
            pw.frame = 100000 * (i + 1)
            sc.render(pw)
        lib.jade_oracle_kind_counters(buf, 1)
        t = time.time()
        rgb, bgr, st = sc.render(p)
        lib.jade_oracle_set_prune(0, 0, 0, 0)
        lib.jade_oracle_kind_counters(buf, 1)
        a = np.array(list(buf), dtype=np.float64).reshape(4, 2, 5)
        same = np.array_equal(rgb.view(np.uint32), rgb0.view(np.uint32))
        tot = a.sum(axis=(0, 1))
        heavy = a[1:].sum(axis=(0, 1))
        print("%-28s frame %s  shadow+env+indirect queries: internal %.2f  leaves %.2f  tests %.2f per query  [%.0f s]" % (
            what, "bit-identical" if same else "DIFFERS", heavy[1] / heavy[0], heavy[2] / heavy[0], heavy[3] / heavy[0], time.time() - t), flush=True)
        for k in range(1, 4):
            for e in range(2):
                n = a[k, e, 0]
                if n:
                    print("    %-12s %-9s %5.1f %% of queries  internal %.1f  leaves %.1f  tests %.1f  cache answers %.1f %%" % (
                        KIND[k], "early" if e else "to the end", 100 * n / tot[0], a[k, e, 1] / n, a[k, e, 2] / n, a[k, e, 3] / n, 100 * a[k, e, 4] / n))
