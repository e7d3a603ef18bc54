#!/usr/bin/env python3
"""Do independent shares of one frame, rendered CONCURRENTLY on one GPU (a host thread and a HIP stream each), fill the
gaps a single pipeline leaves - the drain at the end of every k_trace launch, the thin late passes, the host's waits?

jade_render_multi with the same device listed N times does exactly that (tiles dealt (tx + ty) % N, one thread + stream
per share).  Compared with one share holding the whole frame, same total samples.  GPU box:
    python3 tools/concurrency_test.py [--spp 2048] [--shares 1 2 3 4]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--spp", type=int, default=2048)
    ap.add_argument("--shares", type=int, nargs="+", default=[1, 2, 3, 4])
    ap.add_argument("--state-gb", type=float, default=150.0, help="device memory for path state, split between the shares")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "concurrency.json"))
    a = ap.parse_args()
    import jaderaytracerendering_amd as J
    from jaderaytracerendering_amd import backend as B
    hip = J.hip()
    hs, cfg = J.build_config(a.config)
    res = []
    for n in a.shares:
        scenes = [hip.scene(hs) for _ in range(n)]
        p = B.make_params(cfg.width, cfg.height, a.spp, list(cfg.eye), list(cfg.camera))
        p.max_state_bytes = int(a.state_gb * 1e9 / n)
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            _, _, st = B.render_multi(hip, scenes, p, want_rgb=False, want_bgr8=False) if n > 1 else scenes[0].render(p, want_rgb=False, want_bgr8=False)
            dt = time.perf_counter() - t0
            if best is None or dt < best[0]:
                best = (dt, st)
        dt, st = best
        rays = st.rays_primary + st.rays_secondary
        r = {"shares": n, "seconds": dt, "Mray_per_s": rays / dt / 1e6, "rays": rays, "kernel_ms_max_over_shares": st.kernel_ms,
             "trace_ms_max_over_shares": st.trace_ms, "trace_launches": st.trace_launches}
        print(json.dumps(r), flush=True)
        res.append(r)
        for s in scenes:
            s.close()
    json.dump({"config": a.config, "spp": a.spp, "runs": res}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
