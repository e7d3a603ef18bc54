#!/usr/bin/env python3
"""Development A/B: time 256-spp steps of C3 (AB_CONFIG=C5: 64-spp steps of C5 at 3840x2160; AB_CLOSEUP=1: C3 with the
statue filling the frame) with several builds of libjade_hip*.so in ONE process, variants interleaved
(cdna_hip_programming.md rule 24).  Every variant holds its own path state: the device memory is split between them
through jade_render_params.max_state_bytes, so the records per pixel are fewer than in a bench run (printed).
usage: ab_variants.py "" _A _B ...      ("" = libjade_hip.so; a name may end in @0 / @1 / @2: jade_render_params.walk, default 1 = early exits;
                                         and in %VAR=VAL,...: environment switches read at jade_scene_create, e.g. "_wu@1%JADE_WIDE=1")"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import jaderaytracerendering_amd as J  # noqa: E402
from jaderaytracerendering_amd import _abi, backend as B  # noqa: E402

CONFIG = os.environ.get("AB_CONFIG", "C3")
ROUNDS = int(os.environ.get("AB_ROUNDS", "3"))
W, H, SPP = (3840, 2160, 64) if CONFIG == "C5" else (1920, 1080, 256)
hs, cfg = J.build_config(CONFIG)
eye = list(cfg.eye)
if os.environ.get("AB_CLOSEUP"):
    centre = hs.vertices()[hs.tri_i32()[:, 0] == 0].reshape(-1, 3).mean(0)
    eye = [float(x) for x in centre - 0.22 * (-np.array(cfg.camera[8:11], np.float32))]
names = sys.argv[1:] or [""]
budget = int(float(os.environ.get("AB_STATE_GB", "230")) * 1e9 / len(names))
scenes = {}
for name in names:
    name_, _, envs = name.partition("%")   # "<lib suffix>@<walk>%VAR=VAL,VAR2=VAL": environment switches the module reads at jade_scene_create
    libname, _, walk = name_.partition("@")
    be = B.Backend(os.path.join(ROOT, "jaderaytracerendering_amd", "lib", "libjade_hip%s.so" % libname))
    saved = {}
    for kv in filter(None, envs.split(",")):
        k, _, v = kv.partition("=")
        saved[k] = os.environ.get(k)
        os.environ[k] = v
    sc = be.scene(hs)
    for k, v in saved.items():
        if v is None:
            del os.environ[k]
        else:
            os.environ[k] = v
    p = B.make_params(W, H, SPP * (ROUNDS + 1), eye, list(cfg.camera), walk=int(walk or 1))  # (announced = what will be rendered: the partial sums are sized by it)
    p.max_state_bytes = budget
    sc.begin(p)
    sc.step(SPP)
    scenes[name or "base"] = sc
print("records per pixel:", {n: sc.query(_abi.Q_RECORDS_PER_PIXEL) for n, sc in scenes.items()}, flush=True)
best = {}
for rnd in range(ROUNDS):
    for name, sc in scenes.items():
        st = _abi.Stats()
        t = time.perf_counter()
        sc.step(SPP, st)
        dt = time.perf_counter() - t
        rays_t = st.rays - st.rays_inline - st.rays_tail
        r = (st.rays / dt / 1e6, st.trace_ms, st.light_ms, st.kernel_ms, rays_t / max(st.trace_ms, 1e-9) / 1e3,
             (st.nodes_visited - st.nodes_inline - st.nodes_tail) / max(rays_t, 1), (st.tris_tested - st.tris_inline - st.tris_tail) / max(rays_t, 1))
        if name not in best or r[0] > best[name][0]:
            best[name] = r
for name, r in best.items():
    print("%-8s Mray/s %6.0f  k_trace %6.1f ms (%5.0f Mray/s, V/ray %5.1f T/ray %5.1f)  k_light %6.1f ms  device %6.1f ms" % (name, r[0], r[1], r[4], r[5], r[6], r[2], r[3]), flush=True)
