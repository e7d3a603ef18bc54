#!/usr/bin/env python3
"""bench.py — Mray/s of the jade path-tracing hot path on MI355X.

Contract (driver): python bench.py --gpus N --steps K --warmup W ; for N > 1 the driver launches it with
torch.distributed.run, one rank per GPU (RCCL).  Started by hand with --gpus N > 1 and no launcher, it starts
that launcher itself as a child process — before anything touches the GPU — and exits with its code.  One JSON
line on rank 0.

Workload (BASELINE.json configs[2]/[3]): the 70k-triangle jade statue scene at 1920x1080.  A "step" is one pass of
the hot path adding `spp_per_step` samples to every pixel this rank owns; the per-sample RNG streams and radiance
sums stay on the GPU between steps, so K steps are K*spp_per_step samples of the same render, not K restarts; the
defaults (4 steps x 1024 spp) are exactly the 4096-spp render BASELINE.json names.  Multi-GPU: the image's 16x16
tiles are dealt (tx + ty) % N to the ranks and the samples per step scale with N, so per-GPU work per step is
constant ("weak"); after the timed steps the framebuffer is collected with ONE gather (RCCL), timed separately as
gather_ms.  A step may hand its last few unfinished paths to the next step (jade_render_flush, jade_rt.h); the
warm-up is flushed before the clock starts and the K timed steps are flushed before it stops, so every sample of
the K steps — `samples` = K * spp * pixels, and all their rays — is computed inside the timed region.

value = (primary + secondary rays traced by all ranks in the K timed steps) / max-over-ranks wall time, Mray/s.
        A ray = one hitBVH query (PathTrace.cu:795).
roofline: the dominant kernel, k_trace (BVH traversal + triangle tests), against the arithmetic roof of a kernel
        without matrix work: VALU issue.  achieved = VALU lane-operations per second = (lane-ops per ray,
        SQ_THREAD_CYCLES_VALU from the rocprofv3 --pmc pass of this same command, profiles/valu_issue.json) x (rays
        this run traced) / (k_trace time of this run, HIP events on the kernel's own stream); peak = 1024 SIMDs x 32
        lanes per clock x 2.4 GHz (see VALU_PEAK_TLANEOPS).  frac = (lanes active per VALU instruction / 64) x (VALU
        instructions issued per SIMD per 2 clocks) x (clock held / 2.4 GHz), <= 1.  `binding` says what limits the
        kernel on this configuration: on C3 (scene L2-resident) NO unit is saturated since the wave-wide leaf queue cut
        the instructions per ray by a third - the kernel is bound by the latency of a ray's chain of dependent node
        visits (`sensitivity` carries the ablations, profiles/sensitivity_r02.json); on C5 (873k
        triangles: 51 % L2 hits, 5.9 TB/s) it is HBM.
roofline_hbm: the HBM view SURVEY.md 8d prices the path with.  achieved = MEASURED HBM bytes (PMC FETCH_SIZE x 2 +
        WRITE_SIZE per ray, profiles/hbm_traffic.json) x rays / k_trace time, against 8 TB/s; `algorithmic_GBps` is
        the reference traversal's 40 B per node record + 36 B per triangle test delivered per second — it exceeds
        the HBM peak because the scene is L2-resident, which is why it is not a roofline.
cpu_baseline: the CPU oracle ("port": the reference has no CPU integrator) on a bounded sample of the same scene,
        rank 0 at N = 1 only.
extras (N = 1): secondary rays by call site, and the same scene with the camera moved in until the statue fills
        the frame (every pixel a jade path): the throughput on the rays the headline frame has few of.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s
# VALU issue roof (MI355X_MICROARCH.md: 4 SIMD-32 per CU, a wave64 VALU instruction issues in 2 clocks): 32 lane-operations per
# clock per SIMD.  That rate needs instructions the sequencer can pair (SQ_ACTIVE_INST_VALU2); a wave's own dependent stream issues
# one per 4 clocks, which is the unit SQ_ACTIVE_INST_VALU counts in (1.007 quad-cycles per VALU instruction in every kernel
# here).  The first k_trace of round 2 sat at 4 x ACTIVE_INST_VALU / SIMD-cycles = 1.05 of a possible 2 and its time followed
# its instruction count; the final one issues a third fewer instructions and sits at 0.7.  The roofline below
# is priced against the full 2-clock rate, `issue_busy_of_2` says how far the issue side is from it.
VALU_PEAK_TLANEOPS = 1024 * 32 * 2.4e9 / 1e12  # 256 CUs x 4 SIMDs x 32 lanes, 2.4 GHz = 78.6 T lane-ops/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--spp-per-step", type=int, default=1024, help="samples per pixel per step at N = 1")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the statue-filling camera variant")
    ap.add_argument("--bvh", default="sah", choices=["sah", "lbvh", "ploc"],
                    help="sah: the reference's host builder (default, what the metric is quoted on); lbvh / ploc: GPU builders")
    ap.add_argument("--bvh-leaf", type=int, default=0, help="triangles per leaf for the GPU builders (0: 8 for lbvh as in the reference, 3 for ploc)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) for real runs; gloo rehearses the multi-rank flow with every rank on one GPU")
    ap.add_argument("--virtual-ranks", type=int, default=0,
                    help="development: render rank 0's share of a V-GPU run on this one GPU (partition and spp as at N=V)")
    ap.add_argument("--cpu-spp", type=int, default=8, help="oracle sample: spp over the full frame")
    return ap.parse_args()


def spawn_ranks(n):
    """--gpus N > 1 without a launcher: start torch.distributed.run as a CHILD process (never exec: a process that
    has initialised the GPU must not be replaced, and this one has not touched it yet) and return its exit code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def profile_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    args.gpus = world

    import torch
    import torch.distributed as dist

    import jaderaytracerendering_amd as J
    from jaderaytracerendering_amd import _abi, backend as B, distributed as D, host as H

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
    dev_build_ms = None
    if args.bvh != "sah":
        sb = J.SceneBuilder()
        cfg = sb.config(args.config)
        hs, dev_build_ms = sb.build_device_bvh(hip, args.bvh, leaf_size=args.bvh_leaf or (3 if args.bvh == "ploc" else 8), device_id=local_rank)
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
    if rehearsal and world > 1:  # ranks share one GPU: each may hold its share of the memory, not 60 % of what is free
        params.max_state_bytes = int(0.6 * torch.cuda.mem_get_info(local_rank)[0] / world)
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
    syncs_in_steps = int(st.host_syncs)
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

    cls_keys = ("rays_primary", "rays_shadow", "rays_env", "rays_indirect", "rays_mirror", "rays_refract")
    vals = torch.tensor([float(st.rays_primary + st.rays_secondary), float(st.nodes_visited), float(st.tris_tested),
                         st.trace_ms, float(st.trace_launches), float(st.samples)] + [float(getattr(st, k)) for k in cls_keys],
                        dtype=torch.float64, device=cdev)
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(vals, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    rays_all = float(vals[0].item())

    if rank == 0:
        # rooflines of the dominant kernel (k_trace) on THIS rank
        # k_trace traces what the fused first-pass kernel (k_light: camera rays, floor mirrors) did not trace itself
        rays_rank = float(st.rays_primary + st.rays_secondary - st.rays_inline)
        alg_bytes = 40.0 * st.nodes_visited + 36.0 * st.tris_tested  # (whole step: V and T are not split by kernel)
        launches = max(int(st.trace_launches), 1)
        trace_s = st.trace_ms * 1e-3
        # per-ray counter figures exist for the configurations that were profiled (same scene, frame and BVH)
        key = {"C3": "C3", "C4": "C3", "C5": "C5"}.get(args.config) if (args.bvh == "sah" and not args.width and not args.height) else None
        valu = (profile_json("valu_issue.json") or {}).get(key)
        hbm = (profile_json("hbm_traffic.json") or {}).get(key)
        roof = {"bound": "valu", "kernel": "k_trace", "achieved": None, "peak": VALU_PEAK_TLANEOPS, "unit": "Tlane-op/s", "frac": None,
                "traffic": None, "avg_launch_ms": st.trace_ms / launches, "launches": launches,
                "trace_share_of_step_time": trace_s / (st.kernel_ms * 1e-3) if st.kernel_ms else None,
                # what a profiler sees for the same command: every k_trace launch of the process, warm-up included (launches count
                # the empty ones behind the end of a step's batch too - they are launches, of a few microseconds)
                "launches_incl_warmup": launches + int(st_w.trace_launches),
                "avg_launch_ms_incl_warmup": (st.trace_ms + st_w.trace_ms) / max(launches + int(st_w.trace_launches), 1),
                "rays_per_launch": rays_rank / launches, "rays_traced_by_this_kernel": rays_rank,
                "Mray_per_s_of_this_kernel": rays_rank / trace_s / 1e6 if trace_s > 0 else None}
        if valu and valu.get("valu_lane_ops_per_ray") and trace_s > 0:
            ach = valu["valu_lane_ops_per_ray"] * rays_rank / trace_s / 1e12
            roof.update({"achieved": ach, "frac": ach / VALU_PEAK_TLANEOPS, "issue_busy_of_2": valu.get("valu_busy"),
                         "valu_lane_ops_per_ray": valu["valu_lane_ops_per_ray"], "valu_wave_insts_per_ray": valu.get("valu_wave_insts_per_ray"),
                         "lanes_per_valu_inst_of_64": valu.get("lanes_per_valu_inst"), "valu_busy_profiled": valu.get("valu_busy"),
                         "counters_from": valu.get("source"),
                         "note": "lane-ops per ray from the rocprofv3 --pmc SQ pass of this command (tracked summary named in "
                                 "counters_from); rays and k_trace time are this run's.  frac = share of the chip's VALU lane-slots (32 per "
                                 "SIMD per clock) that carry this kernel's work; issue_busy_of_2 = 4 x SQ_ACTIVE_INST_VALU / SIMD-cycles"})
        roof_hbm = {"bound": "hbm", "kernel": "k_trace", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                    "algorithmic_GBps": alg_bytes / (trace_s + st.light_ms * 1e-3) / 1e9 if trace_s > 0 else None,
                    "algorithmic_bytes_per_launch": alg_bytes / launches,
                    "algorithmic_bytes_per_ray": alg_bytes / float(st.rays_primary + st.rays_secondary),
                    "note_algorithmic": "40 B x V + 36 B x T of ALL rays of the step (k_light's included) over k_trace's time alone would overstate: "
                                        "algorithmic_GBps divides by the device time of both tracing kernels"}
        if hbm and hbm.get("k_trace_hbm_bytes_per_ray") and trace_s > 0:
            b = hbm["k_trace_hbm_bytes_per_ray"] * rays_rank
            roof_hbm.update({"achieved": b / trace_s / 1e9, "frac": b / trace_s / 1e9 / HBM_PEAK_GBS, "traffic": b / launches,
                             "l2_hit_rate_profiled": hbm.get("k_trace_l2_hit_rate"), "counters_from": hbm.get("source"),
                             "note": "measured HBM bytes (2 x FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md) per ray from the --pmc passes "
                                     "of this command; the BVH is cache-resident, so HBM sees the ray records, not the traversal"})
            roof["traffic"] = b / launches
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
            "rays_incl_warmup_this_rank": float(st.rays_primary + st.rays_secondary + st_w.rays_primary + st_w.rays_secondary),
            "rays_k_trace_incl_warmup_this_rank": float(st.rays_primary + st.rays_secondary - st.rays_inline + st_w.rays_primary
                                                        + st_w.rays_secondary - st_w.rays_inline),
            # the step by kernel on this rank: k_light (fused first pass: light samples traced and shaded in one kernel),
            # k_trace (everything else that is traced), the rest = k_shade / k_arm / gaps
            "kernels": {"k_light": {"ms_per_step": st.light_ms / max(args.steps, 1), "rays": float(st.rays_inline),
                                    "Mray_per_s": st.rays_inline / (st.light_ms * 1e-3) / 1e6 if st.light_ms else None},
                        "k_trace": {"ms_per_step": st.trace_ms / max(args.steps, 1), "rays": rays_rank,
                                    "Mray_per_s": rays_rank / trace_s / 1e6 if trace_s > 0 else None},
                        "device_ms_per_step": st.kernel_ms / max(args.steps, 1)},
            "rays_by_call_site": {k[5:]: float(vals[6 + i].item()) for i, k in enumerate(cls_keys)},
            "gather_ms": gather_ms,
            "frame_ok": frame_ok,
            # host waits for the device: per step (the fused first pass, then ONE batch of up to 32 shade / trace passes that
            # stops itself at the carry-over point), and in the flush that finishes the last paths of the render
            "host_syncs_per_step": syncs_in_steps / max(args.steps, 1),
            "host_syncs_in_final_flush": int(st.host_syncs) - syncs_in_steps,
            "scene_build_s": build_s,
            "bvh": args.bvh, "device_bvh_ms": dev_build_ms,
            "nodes_per_ray": float(vals[1].item()) / rays_all, "tris_per_ray": float(vals[2].item()) / rays_all,
            "roofline": roof,
            "roofline_hbm": roof_hbm,
            # HBM when the measured traffic is the larger share of its roof; otherwise neither roof binds
            "binding": None if roof["frac"] is None or roof_hbm["frac"] is None else
                       ("hbm" if roof_hbm["frac"] > roof["frac"] else "latency (no unit saturated: VALU issue %.2f of 2, HBM %.0f %% of peak)"
                        % (roof.get("issue_busy_of_2") or 0.0, 100.0 * roof_hbm["frac"])),
        }
        sens = profile_json("sensitivity_r02.json")
        if sens and key == "C3":
            out["sensitivity"] = {"source": "profiles/sensitivity_r02.json", "k_trace_ms_per_256spp_step": sens.get("ablations_k_trace_ms_per_256spp_step"),
                                  "reading": sens.get("reading")}
        if world == 1 and part_world == 1 and not args.no_extras and args.config in ("C2", "C3", "C4"):
            out["statue_closeup"] = closeup(scene, hip, B, H, _abi, cfg, width, height, args.spp_per_step)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(hs, cfg, width, height, args.cpu_spp)
        print(json.dumps(out), flush=True)
    scene.close()
    if world > 1:
        dist.destroy_process_group()


def closeup(scene, hip, B, H, _abi, cfg, width, height, spp):
    """The same scene with the camera moved in until the statue fills the frame: every pixel starts a jade path
    (BSSRDF / SSS / mirror branches, ~4 shadow + environment + indirect rays per bounce).  The headline frame is
    ~5 % statue; this is the rate on the rays it has few of.  One warm-up step, one timed step, both flushed."""
    import numpy as np
    hs = scene.host_scene
    centre = hs.vertices()[hs.tri_i32()[:, 0] == 0].reshape(-1, 3).mean(0)  # object 0 is the statue (scene_io.cpp add_jade_scene)
    forward = -np.array(cfg.camera[8:11], np.float32)                       # the view axis: M . (0, 0, -1, 0)
    eye = centre - 0.22 * forward                                           # C3 looks at it from 0.56 away
    spp = max(1, min(spp, 256))
    p = B.make_params(width, height, spp, [float(x) for x in eye], list(cfg.camera))
    scene.begin(p)
    w = _abi.Stats()
    scene.step(spp, w)
    scene.flush(w)
    st = _abi.Stats()
    t0 = time.perf_counter()
    scene.step(spp, st)
    scene.flush(st)
    dt = time.perf_counter() - t0
    rays = float(st.rays_primary + st.rays_secondary)
    return {"value": rays / dt / 1e6, "unit": "Mray/s", "spp": spp, "rays_per_sample": rays / max(st.samples, 1),
            "nodes_per_ray": st.nodes_visited / rays, "tris_per_ray": st.tris_tested / rays,
            "k_trace_Mray_per_s": rays / (st.trace_ms * 1e-3) / 1e6 if st.trace_ms else None,
            "trace_share_of_step_time": st.trace_ms / st.kernel_ms if st.kernel_ms else None,
            "camera": "C3's view direction, eye moved to 0.22 from the statue's centre (C3: 0.56)"}


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
