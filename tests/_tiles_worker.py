"""Worker for test_gpu_bench.py::test_device_tiles_feed_the_gather (fresh process: torch initialises HIP first)."""
import os
import sys

import numpy as np
import torch

torch.cuda.init()  # BEFORE libjade_hip.so touches HIP: torch bundles its own runtime, and order matters

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import jaderaytracerendering_amd as J  # noqa: E402
from jaderaytracerendering_amd import backend as B, distributed as D  # noqa: E402

hip = J.hip()
hs, cfg = J.build_config("tinyjade")
w, h = 70, 50
frames = []
with hip.scene(hs) as sc:
    full = None
    for world in (1, 3):
        parts = []
        for r in range(world):
            p = B.params_from_config(cfg, spp=4, tile_rank=r, tile_nranks=world)
            p.width, p.height = w, h
            sc.begin(p)
            sc.step(4)
            n = hip.owned_tile_count(w, h, r, world)
            assert n == len(D.owned_tile_ids(w, h, r, world))
            t = torch.empty((n, D.TILE, D.TILE, 3), dtype=torch.float32, device="cuda")
            sc.resolve_tiles_device(t.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            parts.append(t.cpu().numpy())
            if world == 1:
                full = sc.resolve()[0]
                frames.append(D.gather_framebuffer(t, w, h).cpu().numpy())
        if world > 1:
            out = np.zeros((h, w, 3), np.float32)
            for r in range(world):
                B.assemble_tiles(parts[r], w, h, r, world, out)
            frames.append(out)
for f in frames:
    assert np.array_equal(f.view(np.uint32), full.view(np.uint32))
print("tiles ok")
