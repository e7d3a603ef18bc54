"""One-shot GPU check used during development: python tests/gpu_quick.py [config] [size] [spp]."""
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/tests/", 1)[0])
import jaderaytracerendering_amd as J  # noqa: E402
from jaderaytracerendering_amd import backend as B  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "tiny"
hs, cfg = J.build_config(name)
p = B.params_from_config(cfg)
if len(sys.argv) > 2:
    p.width = p.height = int(sys.argv[2])
if len(sys.argv) > 3:
    p.spp = int(sys.argv[3])
root = __file__.rsplit("/tests/", 1)[0]
orc = B.Backend(root + "/oracle/libjade_oracle.so")
hip = B.hip()
print("devices", hip.device_count(), "scene", name, hs.n_triangles, "tris depth", hs.bvh_depth, flush=True)
with hip.scene(hs) as sh:
    t = time.time(); r_h, b_h, st_h = sh.render(p); dt = time.time() - t
    print("hip   ", round(dt, 3), "s", st_h.as_dict(), "Mray/s(kernel)", st_h.rays / st_h.kernel_ms / 1e3, flush=True)
with orc.scene(hs) as so:
    t = time.time(); r_o, b_o, st_o = so.render(p); dt = time.time() - t
    print("oracle", round(dt, 3), "s", st_o.as_dict(), "Mray/s", st_o.rays / dt / 1e6, flush=True)
keys = ("rays_primary", "rays_secondary", "nodes_visited", "tris_tested", "shaded_hits", "samples")
print("counters equal:", all(getattr(st_h, k) == getattr(st_o, k) for k in keys))
err = np.sqrt(((r_h.astype(np.float64) - r_o) ** 2).sum()) / np.sqrt((r_o.astype(np.float64) ** 2).sum())
print("rel L2", err, "max abs", np.abs(r_h - r_o).max(), "bytes differing", (b_h != b_o).mean())
