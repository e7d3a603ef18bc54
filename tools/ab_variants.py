#!/usr/bin/env python3
"""Development A/B: time one 256-spp step of C3 (AB_CONFIG=C5: one 64-spp step of C5 at 3840x2160) with several builds of libjade_hip*.so in ONE process
(interleaved variants, cdna_hip_programming.md rule 24).  usage: ab_variants.py "" _A _B ..."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jaderaytracerendering_amd as J  # noqa: E402
from jaderaytracerendering_amd import _abi, backend as B  # noqa: E402

# every variant holds its own path state: cap the records so that several fit the device side by side
os.environ.setdefault("JADE_RECORDS_PER_PIXEL", "32")
CONFIG = os.environ.get("AB_CONFIG", "C3")
W, H, SPP = (3840, 2160, 64) if CONFIG == "C5" else (1920, 1080, 256)
hs, cfg = J.build_config(CONFIG)
scenes = {}
for name in sys.argv[1:]:
    be = B.Backend(os.path.join(ROOT, "jaderaytracerendering_amd", "lib", "libjade_hip%s.so" % name))
    sc = be.scene(hs)
    sc.begin(B.make_params(W, H, SPP, list(cfg.eye), list(cfg.camera)))
    sc.step(SPP)
    scenes[name or "base"] = sc
best = {}
for rnd in range(2):
    for name, sc in scenes.items():
        st = _abi.Stats()
        t = time.perf_counter()
        sc.step(SPP, st)
        dt = time.perf_counter() - t
        r = (st.rays / dt / 1e6, st.trace_ms, st.kernel_ms)
        if name not in best or r[0] > best[name][0]:
            best[name] = r
for name, r in best.items():
    print("%-6s Mray/s %.0f  trace_ms %.0f  kernel_ms %.0f" % ((name,) + r), flush=True)
