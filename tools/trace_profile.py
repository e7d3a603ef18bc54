#!/usr/bin/env python3
"""Where do k_trace's clocks go?  

Runs the benchmark frame (C3, 1920x1080) on a -DJADE_TRACE_PROFILE=1 build of the HIP module (make variant NAME=_prof
DEFS=-DJADE_TRACE_PROFILE=1), whose k_trace brackets the pieces of its loop with s_memtime and drains the memory counters at
the end of each piece, and prints / writes the shader clocks per piece summed over all waves, per wave and per unit of work.
The profile build serialises what the product build may overlap (~10 % slower): the laps are a breakdown, not a timing.

usage (GPU box): python3 tools/trace_profile.py [--config C3] [--spp 256] [--closeup] [--out gpurun_out/trace_stalls.json]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LAPS = ["top", "writeback", "refill", "pick", "walk_load", "walk_math", "walk_ring", "test_pop", "test_load", "test_ray", "test_math",
        "test_cand", "resolve"]
COUNTS = ["iterations", "walk_units", "test_units", "resolves", "refills", "waves", "walk_lanes", "test_lanes",
          "walk_wait_lanes", "walk_free_lanes", "test_units_56_or_more_lanes", "test_units_16_or_fewer_lanes", "lanes_of_thin_test_units"]


PK_LAPS = ["advance", "prep", "node", "leaf", "solve", "pop", "fold", "store"]
PK_COUNTS = ["packets", "given_up", "node_visits", "pair_records", "pair_records_with_a_solve", "rays", "node_visit_lanes", "pair_record_lanes", "solve_lanes"]


def packet_summary(hip, st):
    """k_light_packet's laps and counts of the timed steps (jade_debug_packet_profile)."""
    fn = hip.lib.jade_debug_packet_profile
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    buf = (ctypes.c_ulonglong * 64)()
    n = fn(buf, 64, 1)
    assert n == len(PK_LAPS) + len(PK_COUNTS), n
    v = [int(buf[i]) for i in range(n)]
    laps, cnt = dict(zip(PK_LAPS, v[:len(PK_LAPS)])), dict(zip(PK_COUNTS, v[len(PK_LAPS):]))
    total = max(sum(laps.values()), 1)
    pk = max(cnt["packets"], 1)
    return {"first_pass_ms": st.light_ms, "clocks_total_all_waves": total, "counts": cnt, "share": {k: x / total for k, x in laps.items()},
            "per_packet": {"clocks": total / pk, "rays": cnt["rays"] / pk, "node_visits": cnt["node_visits"] / pk, "pair_records": cnt["pair_records"] / pk,
                           "pair_records_with_a_solve": cnt["pair_records_with_a_solve"] / pk, "given_up": cnt["given_up"] / pk},
            "clocks_per": {"node_visit": laps["node"] / max(cnt["node_visits"], 1), "pair_record": laps["leaf"] / max(cnt["pair_records"], 1),
                           "pair_record_with_a_solve": laps["solve"] / max(cnt["pair_records_with_a_solve"], 1),
                           "packet_advance": laps["advance"] / pk, "packet_prep": laps["prep"] / pk, "packet_fold": laps["fold"] / pk},
            "lanes_per": {"node_visit": cnt["node_visit_lanes"] / max(cnt["node_visits"], 1), "pair_record": cnt["pair_record_lanes"] / max(cnt["pair_records"], 1),
                          "solve": cnt["solve_lanes"] / max(cnt["pair_records_with_a_solve"], 1)},
            "note": "shader clocks (s_memtime) summed over waves, 4 waves share a SIMD; 'advance' holds the sample bookkeeping, the camera ray and the mirror bounce, 'fold' the sky look-up and consume_mirror"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="C3")
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--closeup", action="store_true")
    ap.add_argument("--reference-walk", action="store_true", help="jade_render_params.walk = JADE_WALK_REFERENCE (default: early exits, as bench.py)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "trace_stalls.json"))
    a = ap.parse_args()
    os.environ["JADE_HIP_LIB"] = os.path.join(ROOT, "jaderaytracerendering_amd", "lib", "libjade_hip_prof.so")
    import numpy as np
    import jaderaytracerendering_amd as J
    from jaderaytracerendering_amd import _abi, backend as B
    hip = J.hip()
    fn = hip.lib.jade_debug_trace_profile
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    hs, cfg = J.build_config(a.config)
    eye = list(cfg.eye)
    if a.closeup:
        centre = hs.vertices()[hs.tri_i32()[:, 0] == 0].reshape(-1, 3).mean(0)
        forward = -np.array(cfg.camera[8:11], np.float32)
        eye = [float(x) for x in centre - 0.22 * forward]
    p = B.make_params(cfg.width, cfg.height, a.spp * (a.steps + 1), eye, list(cfg.camera), walk=0 if a.reference_walk else 1)
    buf = (ctypes.c_ulonglong * 64)()
    with hip.scene(hs) as sc:
        sc.begin(p)
        w = _abi.Stats()
        sc.step(a.spp, w)
        sc.flush(w)
        n = fn(buf, 64, 1)
        assert n == len(LAPS) + len(COUNTS), n
        if hasattr(hip.lib, "jade_debug_packet_profile"):  # (the warm-up's laps: dropped)
            hip.lib.jade_debug_packet_profile.restype = ctypes.c_int
            hip.lib.jade_debug_packet_profile.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
            hip.lib.jade_debug_packet_profile((ctypes.c_ulonglong * 64)(), 64, 1)
        st = _abi.Stats()
        for _ in range(a.steps):
            sc.step(a.spp, st)
        sc.flush(st)
        fn(buf, 64, 1)
        packet = None
        if hasattr(hip.lib, "jade_debug_packet_profile"):
            packet = packet_summary(hip, st)
    v = [int(buf[i]) for i in range(n)]
    laps = dict(zip(LAPS, v[:len(LAPS)]))
    cnt = dict(zip(COUNTS, v[len(LAPS):]))
    total = sum(laps.values())
    rays = st.rays_primary + st.rays_secondary - st.rays_inline
    out = {"config": a.config, "closeup": a.closeup, "spp_per_step": a.spp, "steps": a.steps, "rays_k_trace": rays,
           "k_trace_ms": st.trace_ms, "launches": st.trace_launches, "Mray_per_s_profile_build": rays / st.trace_ms / 1e3 if st.trace_ms else None,
           "clocks_total_all_waves": total, "counts": cnt,
           "share": {k: x / total for k, x in laps.items()},
           "clocks_per_walk_unit": {k: laps[k] / max(cnt["walk_units"], 1) for k in ("walk_load", "walk_math", "walk_ring")},
           "clocks_per_test_unit": {k: laps[k] / max(cnt["test_units"], 1) for k in ("test_pop", "test_load", "test_ray", "test_math", "test_cand")},
           "clocks_per_resolve": laps["resolve"] / max(cnt["resolves"], 1),
           "clocks_per_refill": (laps["writeback"] + laps["refill"]) / max(cnt["refills"], 1),
           "clocks_per_iteration_top_and_pick": (laps["top"] + laps["pick"]) / max(cnt["iterations"], 1),
           "lanes_per_walk_unit": cnt["walk_lanes"] / max(cnt["walk_units"], 1), "lanes_per_test_unit": cnt["test_lanes"] / max(cnt["test_units"], 1),
           "walk_unit_lanes": {"walking": cnt["walk_lanes"] / max(cnt["walk_units"], 1), "waiting_for_tests": cnt["walk_wait_lanes"] / max(cnt["walk_units"], 1),
                               "without_a_ray": cnt["walk_free_lanes"] / max(cnt["walk_units"], 1)},
           "test_units": {"share_with_56_or_more_lanes": cnt["test_units_56_or_more_lanes"] / max(cnt["test_units"], 1),
                          "share_with_16_or_fewer_lanes": cnt["test_units_16_or_fewer_lanes"] / max(cnt["test_units"], 1),
                          "lanes_per_thin_unit": cnt["lanes_of_thin_test_units"] / max(cnt["test_units_16_or_fewer_lanes"], 1)},
           "wave_units_per_ray": {"walk": cnt["walk_units"] * 64 / max(rays, 1) / 64, "test": cnt["test_units"] / max(rays, 1)},
           "nodes_per_ray": st.nodes_visited / max(st.rays, 1), "tris_per_ray": st.tris_tested / max(st.rays, 1),
           "note": "shader clocks (s_memtime) summed over waves; each lap ends with s_waitcnt vmcnt(0) lgkmcnt(0), so *_load laps are issue -> data in registers; 4 waves share a SIMD: a wave's lap includes the time it waits for the SIMD"}
    out["k_light_packet"] = packet
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
