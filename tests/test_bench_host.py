"""bench.py's host-side pieces that need no GPU."""
import json
import os

from conftest import ROOT

import bench


def test_csrc_hash_follows_code_not_comments():
    """profiles/k_trace_counters.json is stamped with a hash of the device sources; bench.py withholds the roofline fractions
    when the tree's hash differs.  A comment or blank-line edit must not do that, a code edit must."""
    base = bench.csrc_hash()
    plain = lambda path: open(path, errors="replace").read()  # noqa: E731
    commented = bench.csrc_hash(lambda path: "// a new comment\n/* and\n a block */\n" + plain(path).replace("\n", "\n\n   "))
    assert commented == base
    edited = bench.csrc_hash(lambda path: plain(path) + ("\nstatic int jade_extra;\n" if path.endswith("jade_trace.h") else ""))
    assert edited != base


def test_counter_file_belongs_to_the_tree():
    """The committed per-ray counters were cut from this build of csrc/ (tools/summarize_prof.py): the driver's bench line then
    carries roofline fractions instead of nulls."""
    ctr = json.load(open(os.path.join(ROOT, "profiles", "k_trace_counters.json")))
    assert ctr["csrc_sha"] == bench.csrc_hash(), "csrc/ changed since profiles/k_trace_counters.json was cut: re-run tools/r04.sh profile <tag> C5, then tools/summarize_prof.py <tag>"
    for key in ("C3", "C5"):
        e = ctr[key]
        assert e["valu_lane_ops_per_ray"] > 0 and e["l2_requests_per_ray"] > 0 and e["hbm_bytes_per_ray"] > 0 and e["fetch_size_factor"] == 1.0


def test_issue_roof_is_the_calibrated_one():
    """bench.py's `valu_issue` roof (one wave64 instruction per SIMD per 4 clocks) is a measurement, not a reading of the guide:
    profiles/valu_calibration.json holds tools/calib/valu_calib's runs - every instruction kind but unpacked fp32 add / mul / fma and
    v_mov stays at 0.25 per clock from 2 waves per SIMD up, those four reach nearly twice that."""
    cal = json.load(open(os.path.join(ROOT, "profiles", "valu_calibration.json")))
    assert abs(cal["single_rate_insts_per_simd_per_clock_5_waves"] - 0.25) < 0.01
    assert abs(bench.VALU_ISSUE_PEAK_GINST - 1024 * 0.25 * 2.4) < 1e-9
    by = {(r["kernel"], r["waves_per_simd"]): r for r in cal["runs"]}
    for kind in ("v_pk_fma_f32", "v_pk_add_f32", "v_max3_f32", "v_cmp_lt_f32+v_cndmask_b32", "v_min_f32+v_cndmask_b32+v_add_u32+v_max_f32"):
        for w in (2, 5, 8):
            assert 0.21 < by[(kind, w)]["wave_insts_per_simd_per_busy_clock"] < 0.27, (kind, w)
    for kind in ("v_fma_f32", "v_add_f32", "v_mul_f32", "v_mov_b32"):
        assert by[(kind, 8)]["wave_insts_per_simd_per_busy_clock"] > 0.45, kind
    # valu_busy (tools/summarize_prof.py) is 4 x instructions per SIMD per clock: 1.0 at this roof
    assert all(abs(r["valu_busy"] - 4 * r["wave_insts_per_simd_per_busy_clock"]) < 0.01 for r in cal["runs"])


# ---- the two ways bench.py is started on more than one GPU -----------------------------------------------
# The driver starts it under `python -m torch.distributed.run` (one rank per GPU) and sets nothing in the environment; started by
# hand with --gpus N it spawns that launcher itself.  Either way every rank must (a) have HSA_ENABLE_IPC_MODE_LEGACY=0 in its
# environment before torch / HIP are loaded (RCCL's bootstrap fails without it on this pool's host driver), (b) rendezvous on
# 127.0.0.1 and (c) own the tiles (tx + ty) % N == rank with N x 1024 samples per step.  JADE_BENCH_PLUMBING_ONLY makes a rank
# print that plumbing and return before it imports torch: no GPU is needed, so the launch forms are covered here.
import socket  # noqa: E402
import subprocess  # noqa: E402
import sys  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ranks(cmd, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("HSA_ENABLE_IPC_MODE_LEGACY", "MASTER_ADDR", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["JADE_BENCH_PLUMBING_ONLY"] = "1"
    env.update(extra_env or {})
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    if out.returncode != 0 and "--master-port" in cmd:  # (a port found free a moment ago may have been taken since: once more, on another)
        cmd = list(cmd)
        cmd[cmd.index("--master-port") + 1] = str(_free_port())
        out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    # (the ranks share one stdout: two of them may print into the same line - scan for objects rather than for lines)
    recs, dec, text, at = [], json.JSONDecoder(), out.stdout, 0
    while (at := text.find("{", at)) >= 0:
        try:
            obj, end = dec.raw_decode(text, at)
        except json.JSONDecodeError:
            at += 1
            continue
        at = end
        if isinstance(obj, dict) and "rank" in obj:
            recs.append(obj)
    return sorted(recs, key=lambda r: r["rank"])


def _check_world(recs, world):
    from jaderaytracerendering_amd import distributed as D
    assert [r["rank"] for r in recs] == list(range(world))
    tx, ty = D.tile_grid(1920, 1080)
    assert sum(r["owned_tiles"] for r in recs) == tx * ty
    for r in recs:
        assert r["world"] == world and r["gpus"] == world and r["local_rank"] == r["rank"]
        assert r["HSA_ENABLE_IPC_MODE_LEGACY"] == ("0" if world > 1 else None)
        assert r["spp_per_step"] == 1024 * world and r["steps"] == 3 and r["warmup"] == 1
        assert not r["torch_loaded"], "the environment must be prepared before torch is imported"
        ids = D.owned_tile_ids(1920, 1080, r["rank"], world)
        assert r["owned_tiles"] == len(ids) and r["first_tiles"] == [int(t) for t in ids[:4]]
        if world > 1:
            assert r["MASTER_ADDR"] == "127.0.0.1"


def test_driver_style_launch_prepares_every_rank():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2 ...`: the driver's form."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    _check_world(_ranks(cmd), 2)


def test_self_spawned_launch_equals_the_driver_style_one():
    """`python bench.py --gpus 2` with no launcher spawns the same launcher as a child: the same ranks, tiles and environment."""
    a = _ranks([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"])
    _check_world(a, 2)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
    b = _ranks(cmd)
    drop = lambda r: {k: v for k, v in r.items() if k != "MASTER_ADDR"}  # noqa: E731
    assert [drop(r) for r in a] == [drop(r) for r in b]


def test_one_rank_under_the_launcher_equals_the_plain_launch():
    """--gpus 1 started under torch.distributed.run (WORLD_SIZE=1) is the plain `python bench.py`: same tiles, same samples, and
    no multi-rank environment is forced on it."""
    plain = _ranks([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1"]
    launched = _ranks(cmd)
    _check_world(plain, 1)
    _check_world(launched, 1)
    drop = lambda r: {k: v for k, v in r.items() if k != "MASTER_ADDR"}  # noqa: E731
    assert [drop(r) for r in plain] == [drop(r) for r in launched]


def test_an_operators_own_ipc_setting_is_kept():
    recs = _ranks([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], {"HSA_ENABLE_IPC_MODE_LEGACY": "1"})
    assert [r["HSA_ENABLE_IPC_MODE_LEGACY"] for r in recs] == ["1", "1"]
