"""The driver-facing pieces on the GPU: device-resident tiles for the gather, and bench.py's JSON line."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_device_tiles_feed_the_gather():
    """jade_render_resolve_tiles_device -> torch tensor -> gather_framebuffer == jade_render_resolve.
    Runs in a fresh process: PyTorch bundles its own HIP runtime and must initialise it BEFORE
    libjade_hip.so makes its first HIP call (the other order leaves torch without a GPU)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_tiles_worker.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "tiles ok" in r.stdout, r.stderr[-2000:]


def test_bench_json_contract():
    """One tiny bench run: the single JSON line the driver parses, with roofline and cpu_baseline objects."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "tinyjade", "--width", "96", "--height", "64",
           "--spp-per-step", "16", "--steps", "2", "--warmup", "1", "--cpu-spp", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Mray/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["frame_ok"] is True
    rf, rs = d["roofline"], d["rooflines"]
    # three roofs for the dominant kernel (VALU lane-ops, L2 requests, HBM bytes); `roofline` is the binding one.  The per-ray
    # counter figures exist for the profiled configurations (C3 / C5 at their own frame size) and for the build they were
    # cut from only, so a tiny run carries the live quantities and null for the rest
    assert set(rs) == {"valu", "valu_issue", "l2", "hbm"} and rf["bound"] in rs and rf["kernel"] == "k_trace"
    assert rs["valu"]["unit"] == "Tlane-op/s" and abs(rs["valu"]["peak"] - 78.6432) < 1e-6
    assert rs["l2"]["unit"] == "GB/s" and rs["l2"]["peak"] == 16800.0 and rs["hbm"]["peak"] == 8000.0
    assert rf["launches"] >= 1 and rf["avg_launch_ms"] > 0 and rf["achieved"] is None and rf["frac"] is None and d["binding"] is None
    assert rf["algorithmic_GBps"] > 0 and rf["tree_build"]
    kt = d["kernels"]["k_trace"]
    assert kt["nodes_per_ray"] > 0 and kt["tris_per_ray"] >= 0 and kt["algorithmic_bytes_per_ray"] > 0
    # the timed region runs with early exits (jade_rt.h, JADE_WALK_EARLY_EXIT); the same steps with the reference's walk are
    # measured beside it, and the algorithmic bytes are those of the reference's walk, not of what was read
    # (the side run has the timed region's own step and warm-up counts)
    rw = d["reference_walk"]
    assert d["config"]["walk"] == "early_exit" and rw["value"] > 0 and rw["steps"] == 2 and rw["warmup"] == 1 and d["value_reference_walk"] == rw["value"]
    assert d["value_early_exit_walk"] == d["value"] and d["early_exit_walk"] is None and d["occluder_cache"] is None
    assert rw["nodes_per_ray_k_trace"] >= kt["nodes_per_ray"] and rw["tris_per_ray_k_trace"] >= kt["tris_per_ray"]
    assert abs(kt["algorithmic_bytes_per_ray"] - (40 * rw["nodes_per_ray_k_trace"] + 36 * rw["tris_per_ray_k_trace"])) < 1e-6 * kt["algorithmic_bytes_per_ray"]
    # the frame the run rendered, against the oracle on a few tiles at the full sample count
    pc = d["parity_check"]
    assert pc["ok"] is True and pc["rel_l2"] <= 1e-4 and pc["bgr_max"] <= 1 and pc["spp"] == 3 * 16 and len(pc["tiles"]) >= 3
    assert d["state"]["records_per_pixel"] >= 1 and d["state"]["sum_lanes"] == 64
    assert set(d["rays_by_call_site"]) == {"primary", "shadow", "env", "indirect", "mirror", "refract"}
    assert sum(d["rays_by_call_site"].values()) == d["rays"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "Mray/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    # samples rendered in the timed region: 2 steps x 16 spp x 96 x 64 pixels
    assert d["samples"] == 2 * 16 * 96 * 64


def test_bench_with_the_occluder_cache():
    """bench.py --walk cached: JADE_WALK_EARLY_EXIT_CACHED in the timed region, both other walks beside it; the frame is checked
    against the oracle like any other (parity_check)."""
    d = _bench_line(["--config", "tinyjade", "--width", "96", "--height", "64", "--spp-per-step", "16", "--steps", "2", "--warmup", "1",
                     "--cpu-spp", "1", "--walk", "cached"])
    rw, ew, oc = d["reference_walk"], d["early_exit_walk"], d["occluder_cache"]
    assert d["config"]["walk"] == "cached" and rw["value"] > 0 and ew["value"] > 0 and d["value_early_exit_walk"] == ew["value"]
    assert rw["nodes_per_ray_k_trace"] >= ew["nodes_per_ray_k_trace"] and rw["tris_per_ray_k_trace"] >= ew["tris_per_ray_k_trace"]
    assert 0 <= oc["answered"] <= d["rays_by_call_site"]["shadow"] + d["rays_by_call_site"]["env"] and 0 <= oc["share_of_shadow_and_env_rays"] <= 1
    assert d["parity_check"]["ok"] is True


def _bench_line(args, timeout=900, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       env=dict(os.environ, **env) if env else None)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_four_rank_rehearsal_gathers_the_one_rank_frame():
    """The multi-rank flow of bench.py at the benchmark's frame size, rehearsed on the one GPU: four rank processes (the
    box admits at most six processes on its card, and this test session is one of them; the driver's real run is eight
    ranks, one per GPU, over RCCL), tiles dealt (tx + ty) % 4, each rank holding a quarter of the memory, the frame
    gathered through torch.distributed (gloo here).  Samples per step scale with the ranks (weak scaling), so four
    ranks x 12 spp = one rank x 48 spp: the gathered frame must be the one-rank frame bit for bit, and so must the rays."""
    common = ["--config", "C3", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"]
    one = _bench_line(common + ["--spp-per-step", "48"])
    six = _bench_line(common + ["--gpus", "4", "--dist-backend", "gloo", "--spp-per-step", "12"])
    assert one["n_gpus"] == 1 and six["n_gpus"] == 4 and six["rehearsal_all_ranks_on_one_gpu"] is True
    assert one["config"]["spp_per_step"] == six["config"]["spp_per_step"] == 48
    assert one["frame_ok"] and six["frame_ok"] and one["frame_sha256"] and one["frame_sha256"] == six["frame_sha256"]
    assert one["rays"] == six["rays"] and one["samples"] == six["samples"] == 48 * 1920 * 1080
    assert one["rays_by_call_site"] == six["rays_by_call_site"]


def test_a_failed_rccl_bring_up_costs_the_gather_its_backend_not_the_bench_its_line():
    """bench.py's control plane (barriers, the reduction of the ranks' counts) runs over gloo; RCCL carries only the frame gather and
    is brought up and exercised before any clock runs.  Here it CANNOT come up - two ranks on the one GPU, which RCCL refuses - and
    the run must go on: every rank agrees to gather through host memory, the line says why, and frame and rays are those of the one-rank run."""
    common = ["--config", "C3", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"]
    one = _bench_line(common + ["--spp-per-step", "16"])
    two = _bench_line(common + ["--gpus", "2", "--dist-backend", "gloo", "--spp-per-step", "8"], env={"JADE_BENCH_TRY_RCCL": "1"})
    assert one["exchange"] is None
    ex = two["exchange"]
    assert ex["backend"].startswith("gloo") and ex["rccl_error"], ex
    assert two["n_gpus"] == 2 and two["frame_ok"] and two["frame_sha256"] == one["frame_sha256"] and two["rays"] == one["rays"]
