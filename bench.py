#!/usr/bin/env python3
"""bench.py — Mray/s of the jade path-tracing hot path on MI355X.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N > 1 it is
launched by torch.distributed.run, one rank per GPU (RCCL).  One JSON line on rank 0.

Workload (BASELINE.json configs[2]/[3]): the 70k-triangle jade statue scene at
1920x1080.  A "step" is one pass of the hot path adding `spp_per_step` samples to
every pixel this rank owns; the per-pixel RNG streams and radiance sums stay on
the GPU between steps, so K steps are K*spp_per_step samples of the same
render, not K restarts; the defaults (4 steps x 1024 spp) are exactly the
4096-spp render BASELINE.json names.  Multi-GPU: the image's 16x16 tiles are dealt
round-robin to the ranks and the samples per step scale with N, so per-GPU work
per step is constant ("weak"); after the timed steps the framebuffer is
collected with ONE gather (RCCL), timed separately as gather_ms.  A step may hand
its last few unfinished paths to the next step (jade_render_flush, jade_rt.h); the
warm-up is flushed before the clock starts and the K timed steps are flushed before
it stops, so every sample of the K steps — `samples` = K * spp * pixels, and all
their rays — is computed inside the timed region.

value = (primary + secondary rays traced by all ranks in the K timed steps)
        / max-over-ranks wall time, in Mray/s.  A ray = one hitBVH query.
roofline: k_trace's algorithmic bytes (40 B per node record needed + 36 B per
        triangle tested) / k_trace time measured with HIP events
        on its own stream (jade_stats.trace_ms), against 8 TB/s HBM.  The bytes are
        algorithmic: the scene is L2-resident, `traffic` (HBM bytes per launch from
        profiles/hbm_traffic.json, PMC) is what HBM really moves.
cpu_baseline: the CPU oracle ("port": the reference has no CPU integrator) on a
        bounded sample of the same scene, rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--spp-per-step", type=int, default=1024, help="samples per pixel per step at N = 1")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--bvh", default="sah", choices=["sah", "lbvh"],
                    help="sah: the reference's host builder (default, what the metric is quoted on); lbvh: GPU builder")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) for real runs; gloo rehearses the multi-rank flow with every rank on one GPU")
    ap.add_argument("--virtual-ranks", type=int, default=0,
                    help="development: render rank 0's share of a V-GPU run on this one GPU (partition and spp as at N=V)")
    ap.add_argument("--cpu-spp", type=int, default=8, help="oracle sample: spp over the full frame")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    import torch
    import torch.distributed as dist

    import jaderaytracerendering_amd as J
    from jaderaytracerendering_amd import _abi, backend as B, distributed as D

    hip = J.hip()  # raises if the HIP extension is missing: no fallback
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    rehearsal = args.dist_backend == "gloo"
    if rehearsal:
        local_rank = 0  # every rank shares GPU 0; collectives go through host memory
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = torch.device("cpu") if rehearsal else dev  # where collective payloads live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    t0 = time.time()
    lbvh_ms = None
    if args.bvh == "lbvh":
        sb = J.SceneBuilder()
        cfg = sb.config(args.config)
        hs, lbvh_ms = sb.build_lbvh(hip, device_id=local_rank if args.dist_backend != "gloo" else 0)
        sb.close()
    else:
        hs, cfg = J.build_config(args.config)
    build_s = time.time() - t0
    width = args.width or cfg.width
    height = args.height or cfg.height
    part_world, part_rank = world, rank
    if args.virtual_ranks > 1 and world == 1:
        part_world, part_rank = args.virtual_ranks, 0
    spp_step = args.spp_per_step * part_world  # weak scaling: fixed work per GPU per step
    params = B.make_params(width, height, spp_step, list(cfg.eye), list(cfg.camera), tile_rank=part_rank,
                           tile_nranks=part_world, device_id=local_rank)
    scene = hip.scene(hs, device_id=local_rank)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    scene.begin(params)
    st_w = _abi.Stats()
    for _ in range(args.warmup):
        scene.step(spp_step, st_w)
    scene.flush(st_w)  # the warm-up's last paths finish outside the timed region ...
    st = _abi.Stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        scene.step(spp_step, st)  # synchronous; a step may hand its last few paths to the next one (jade_rt.h)
    scene.flush(st)  # ... and the timed steps' own inside it: every sample of the K steps is done before the clock stops
    barrier()
    dt = time.perf_counter() - t0

    # the single exchange step: gather the framebuffer on rank 0
    n_owned = hip.owned_tile_count(width, height, part_rank, part_world)
    tiles = torch.empty((n_owned, D.TILE, D.TILE, 3), dtype=torch.float32, device=dev)
    barrier()
    g0 = time.perf_counter()
    scene.resolve_tiles_device(tiles.data_ptr(), torch.cuda.current_stream().cuda_stream)
    frame = D.gather_framebuffer(tiles.to(cdev), width, height) if world > 1 else None
    barrier()
    gather_ms = (time.perf_counter() - g0) * 1e3
    if world == 1 and part_world == 1:
        frame = D.gather_framebuffer(tiles, width, height)
    frame_ok = True
    if rank == 0 and frame is not None:
        frame_ok = bool(torch.isfinite(frame).all().item()) and tuple(frame.shape) == (height, width, 3)

    vals = torch.tensor([float(st.rays_primary + st.rays_secondary), float(st.nodes_visited), float(st.tris_tested),
                         st.trace_ms, float(st.trace_launches), float(st.samples)], dtype=torch.float64, device=cdev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(vals, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    rays_all = float(vals[0].item())

    if rank == 0:
        # roofline of the dominant kernel (k_trace) on THIS rank
        alg_bytes = 40.0 * st.nodes_visited + 36.0 * st.tris_tested
        launches = max(int(st.trace_launches), 1)
        trace_s = st.trace_ms * 1e-3
        achieved = alg_bytes / trace_s / 1e9 if trace_s > 0 else 0.0
        traffic = None
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(prof):
            try:  # PMC: HBM bytes per algorithmic byte of k_trace (tools/summarize_pmc.py), per launch like `achieved`
                traffic = json.load(open(prof)).get("k_trace_hbm_bytes_per_algorithmic_byte") * alg_bytes / launches
            except Exception:
                traffic = None
        out = {
            "metric": "Mray/s (primary+secondary)",
            "value": rays_all / dt / 1e6,
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.config}: jade statue stand-in ({hs.n_triangles:,} triangles, {args.bvh.upper()} BVH), {width}x{height}, "
                            f"{spp_step} spp per step, tiles dealt over {world} GPU(s)",
                "spp_per_step": spp_step, "width": width, "height": height, "triangles": hs.n_triangles,
                "bvh_nodes": hs.n_nodes, "bvh_depth": hs.bvh_depth, "parallelism": f"tiles{world}",
            },
            "virtual_ranks": part_world if part_world != world else None,
            "rehearsal_all_ranks_on_one_gpu": True if rehearsal else None,
            "rays": rays_all,
            "samples": float(vals[5].item()),
            "gather_ms": gather_ms,
            "frame_ok": frame_ok,
            "scene_build_s": build_s,
            "bvh": args.bvh, "lbvh_device_ms": lbvh_ms,
            "nodes_per_ray": float(vals[1].item()) / rays_all, "tris_per_ray": float(vals[2].item()) / rays_all,
            "roofline": {
                "bound": "hbm", "kernel": "k_trace", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes / launches, "avg_launch_ms": st.trace_ms / launches,
                "launches": launches, "trace_share_of_step_time": trace_s / (st.kernel_ms * 1e-3) if st.kernel_ms else None,
                # what a profiler sees for the same command: every k_trace launch of the process, warm-up included
                # (the warm-up closes with its own tail of small launches, so its average is lower)
                "launches_incl_warmup": launches + int(st_w.trace_launches),
                "avg_launch_ms_incl_warmup": (st.trace_ms + st_w.trace_ms) / max(launches + int(st_w.trace_launches), 1),
                "note": "achieved counts the reference traversal's ALGORITHMIC bytes (SURVEY.md 8d); the scene is L2-resident, so "
                        "traffic (HBM bytes per launch, PMC) is ~10x smaller and frac may exceed 1; PMC shows k_trace bound by VALU "
                        "issue (profiles/r01_v7_sq_summary.json, DESIGN.md 3.4)",
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(hs, cfg, width, height, args.cpu_spp)
        print(json.dumps(out), flush=True)
    scene.close()
    if world > 1:
        dist.destroy_process_group()


def usable_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    cores = min(cores, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    cores = min(cores, max(1, q // per))
            break
        except Exception:
            continue
    return cores


def cpu_baseline(hs, cfg, width, height, spp):
    """The oracle (a port: the reference has no CPU integrator) on all host cores."""
    from jaderaytracerendering_amd import backend as B
    lib = os.path.join(ROOT, "oracle", "libjade_oracle.so")
    if not os.path.exists(lib):
        return None
    cores = usable_cores()
    oracle = B.Backend(lib)
    p = B.make_params(width, height, spp, list(cfg.eye), list(cfg.camera), threads=cores)
    with oracle.scene(hs) as so:
        t0 = time.perf_counter()
        _, _, st = so.render(p, want_rgb=False, want_bgr8=False)
        dt = time.perf_counter() - t0
    return {"value": st.rays / dt / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
            "sample": f"same scene and camera, full {width}x{height} frame, {spp} spp ({st.rays} rays, {dt:.1f} s)"}


if __name__ == "__main__":
    main()
