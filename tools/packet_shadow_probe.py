#!/usr/bin/env python3
"""Would k_trace's shadow rays pay as packets, once grouped by the triangle they leave?

Synthetic shadow rays of the jade scene as the BSSRDF / SSS branches make them (PathTrace.cu:931-1178): origin = a random
point of a statue triangle, target = a random point of an emitter triangle, direction unnormalised, the source triangle
skipped.  Three orders of the same rays: grouped by (emitter, origin triangle) - what a sort by the source triangle would
give -, grouped by coarse buckets of 16 triangles, and shuffled (what the queue holds today).  Each order through
jade_trace_rays (k_trace: one lane walks one ray) and through the packet walk (64 rays of a wave together).  GPU box."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jaderaytracerendering_amd as J  # noqa: E402
from jaderaytracerendering_amd import _abi  # noqa: E402

cfgname = sys.argv[1] if len(sys.argv) > 1 else "C3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4_000_000
hs, cfg = J.build_config(cfgname)
hip = J.hip()
fn = hip.lib.jade_debug_packet_rays
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 9
rng = np.random.default_rng(1)
v = hs.vertices()                         # [nT, 3, 3], BVH order
obj = hs.tri_i32()[:, 0]
statue = np.nonzero(obj == 0)[0]
emit = hs.a["emit"].astype(np.int64)
tri = np.sort(rng.choice(statue, n))      # grouped by source triangle (BVH order = spatial order)
e = emit[rng.integers(0, len(emit), n)]


def point_on(t):
    a, b = rng.random(len(t), dtype=np.float32), rng.random(len(t), dtype=np.float32)
    f = a + b > 1
    a[f], b[f] = 1 - a[f], 1 - b[f]
    p = v[t]
    return p[:, 0] + (p[:, 1] - p[:, 0]) * a[:, None] + (p[:, 2] - p[:, 0]) * b[:, None]


o = point_on(tri).astype(np.float32)
d = (point_on(e) - o).astype(np.float32)
skip = tri.astype(np.int32)
orders = {"by (emitter, triangle)": np.lexsort((tri, e)), "by (emitter, triangle >> 4)": np.lexsort((rng.random(n), tri >> 4, e)), "shuffled": rng.permutation(n)}
out = {"config": cfgname, "rays": n, "orders": {}}
with hip.scene(hs) as sc:
    for name, idx in orders.items():
        oo, dd, ss = np.ascontiguousarray(o[idx]), np.ascontiguousarray(d[idx]), np.ascontiguousarray(skip[idx])
        best_t = None
        for rep in range(2):
            i1, t1, p1, st = sc.trace_rays(oo, dd, ss)
            best_t = st.trace_ms if best_t is None else min(best_t, st.trace_ms)
        hit = np.zeros(n, np.int32); dist = np.zeros(n, np.float32); pt = np.zeros((n, 3), np.float32); V = np.zeros(n, np.uint32); T = np.zeros(n, np.uint32)
        best_p = None
        for rep in range(2):
            ms = C.c_double(0)
            hip.check(fn(sc._h, n, oo.ctypes.data, dd.ctypes.data, ss.ctypes.data, hit.ctypes.data, dist.ctypes.data, pt.ctypes.data, V.ctypes.data, T.ctypes.data, C.byref(ms)))
            best_p = ms.value if best_p is None else min(best_p, ms.value)
        ok = bool(np.array_equal(hit, i1) and np.array_equal(dist.view(np.uint32), t1.view(np.uint32)) and int(V.sum()) == st.nodes_visited and int(T.sum()) == st.tris_tested)
        r = {"k_trace_ms": best_t, "k_trace_Mray_per_s": n / best_t / 1e3, "packet_ms": best_p, "packet_Mray_per_s": n / best_p / 1e3, "V_per_ray": st.nodes_visited / n,
             "T_per_ray": st.tris_tested / n, "same_results": ok}
        out["orders"][name] = r
        print(name, json.dumps(r), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "packet_shadow_probe_%s.json" % cfgname), "w"), indent=1)
