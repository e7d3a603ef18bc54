#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz: inputs (the arrays crossing jade_rt.h) + the oracle's outputs.

The reference holds no golden vectors and cannot run here, so these
fixtures pin the oracle and the host pipeline against THEMSELVES across rounds: a change in
either shows up as a diff of committed data.  Run from the repo root after `make host oracle`:
    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jaderaytracerendering_amd as J  # noqa: E402
from jaderaytracerendering_amd import backend as B  # noqa: E402

CASES = {"tiny": dict(), "tinyjade": dict(), "tinyjade_wide": dict(config="tinyjade", width=45, height=27, spp=3)}


def main():
    oracle = B.Backend(os.path.join(ROOT, "oracle", "libjade_oracle.so"))
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, kw in CASES.items():
        hs, cfg = J.build_config(kw.get("config", name))
        p = B.params_from_config(cfg, spp=kw.get("spp", cfg.spp), threads=1)
        p.width, p.height = kw.get("width", cfg.width), kw.get("height", cfg.height)
        with oracle.scene(hs) as sc:
            rgb, bgr, st = sc.render(p)
            rng = np.random.default_rng(42)
            v = hs.vertices().reshape(-1, 3)
            o = (v.mean(0) + (rng.random((256, 3)) - 0.5) * np.ptp(v, axis=0).max() * 1.5).astype(np.float32)
            d = rng.normal(size=(256, 3)).astype(np.float32)
            skip = rng.integers(-1, hs.n_triangles, 256).astype(np.int32)
            ti, td, tp, tst = sc.trace_rays(o, d, skip)
        ctr = np.array([st.rays_primary, st.rays_secondary, st.nodes_visited, st.tris_tested, st.shaded_hits, st.samples,
                        st.rays_shadow, st.rays_env, st.rays_indirect, st.rays_mirror, st.rays_refract], np.uint64)
        hs.save_npz(os.path.join(out_dir, name + ".npz"), rgb=rgb, bgr8=bgr, counters=ctr,
                    params=np.array([p.width, p.height, p.spp, p.frame], np.int64), eye=np.array(p.eye[:], np.float32),
                    camera=np.array(p.camera[:], np.float32), ray_o=o, ray_d=d, ray_skip=skip, ray_hit=ti, ray_dist=td,
                    ray_point=tp, ray_counters=np.array([tst.nodes_visited, tst.tris_tested], np.uint64))
        print(name, os.path.getsize(os.path.join(out_dir, name + ".npz")), "bytes", dict(zip("prim sec V T H n".split(), ctr)))


if __name__ == "__main__":
    main()
