import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import jaderaytracerendering_amd as J
from jaderaytracerendering_amd import backend as B
hip = J.hip()
b = J.SceneBuilder(); cfg = b.config("C2")
sah = b.build()
rng = np.random.default_rng(11); n = 200000
statue = sah.vertices()[sah.tri_i32()[:, 0] == 0].reshape(-1, 3)
ctr, ext = statue.mean(0), np.ptp(statue, axis=0).max()
o = (ctr + (rng.random((n, 3)) - 0.5) * ext * 3).astype(np.float32); d = rng.normal(size=(n, 3)).astype(np.float32); skip = np.full(n, -1, np.int32)
def cost(hs):
    with hip.scene(hs) as sc:
        _, _, _, st = sc.trace_rays(o, d, skip)
    return st.nodes_visited / n, st.tris_tested / n, st.kernel_ms
print("sah", cost(sah), sah.n_nodes)
for kind in ("lbvh", "ploc"):
    for ls in (2, 3, 4, 6, 8):
        hs, ms = b.build_device_bvh(hip, kind, leaf_size=ls)
        v, t, kms = cost(hs)
        print(kind, ls, "V %.1f T %.1f cost %.0f  trace %.2f ms build %.2f ms nodes %d depth %d" % (v, t, 35 * v + 85 * t, kms, ms, hs.n_nodes, hs.bvh_depth))
