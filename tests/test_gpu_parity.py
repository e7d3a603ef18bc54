"""GPU parity: libjade_hip.so (through the C ABI) against the CPU oracle.

The bar (BASELINE.json north_star / SURVEY.md §8c-d):
  * integer work counters (rays, node records V, triangle tests T, shaded
    vertices H, samples) EXACTLY equal: every random draw, hit/miss and branch
    decision of every path is the same;
  * hit triangle indices of raw ray queries bit-exact, hit points bit-exact;
  * pre-tonemap float RGB within 1e-4 relative L2 (the only licensed
    difference: the HIP path sums radiance forward instead of unwinding the
    reference's (dir, rate) stacks, jade_shade.h);
  * BGR8 bytes: at most 1 code value apart, on a vanishing fraction of bytes.
"""
import numpy as np
import pytest

from conftest import assert_cached_walk_equals_reference_walk, B, J, assert_early_exit_equals_reference_walk, config_scene, counters, rel_l2
from jaderaytracerendering_amd import _abi

pytestmark = pytest.mark.gpu

TOL = 1e-4  # relative L2 on pre-tonemap radiance, from BASELINE.json north_star


def _with_walk(params, walk):
    q = type(params).from_buffer_copy(params)
    q.walk = walk
    return q


def _render_both(oracle, hip, hs, params):
    """The frame through the oracle and through the HIP module with the reference's walk (what the callers compare, counters
    included); and, checked right here for every scene of this file, through the HIP module with early exits
    (jade_rt.h, JADE_WALK_EARLY_EXIT): the same bits, the same rays, fewer node records and triangle tests."""
    assert params.walk == _abi.WALK_REFERENCE
    early = _with_walk(params, _abi.WALK_EARLY_EXIT)
    with oracle.scene(hs) as so, hip.scene(hs) as sh:
        r_o, b_o, st_o = so.render(params)
        r_h, b_h, st_h = sh.render(params)
        assert_early_exit_equals_reference_walk((r_h, b_h, st_h), sh.render(early))
        assert_cached_walk_equals_reference_walk(sh, params, (r_h, b_h, st_h))  # ... and with the occluder cache, cold and warm
    # ... and with the wide form of k_trace's walk forced (by default only trees that do not fit the L2 get it)
    import os
    before = os.environ.get("JADE_WIDE")
    os.environ["JADE_WIDE"] = "1"
    try:
        with hip.scene(hs) as sw:
            assert_early_exit_equals_reference_walk((r_h, b_h, st_h), sw.render(early))
            assert_cached_walk_equals_reference_walk(sw, params, (r_h, b_h, st_h))
    finally:
        if before is None:
            del os.environ["JADE_WIDE"]
        else:
            os.environ["JADE_WIDE"] = before
    return (r_o, b_o, st_o), (r_h, b_h, st_h)


def _assert_parity(o, h, tol=TOL):
    (r_o, b_o, st_o), (r_h, b_h, st_h) = o, h
    assert counters(st_h) == counters(st_o)
    assert np.isnan(r_h).sum() == np.isnan(r_o).sum()
    fin = np.isfinite(r_o) & np.isfinite(r_h)
    err = rel_l2(r_h[fin], r_o[fin])
    assert err <= tol, f"relative L2 {err:g} > {tol:g}"
    diff = np.abs(b_h.astype(np.int16) - b_o.astype(np.int16))
    assert diff.max() <= 1
    assert (diff != 0).mean() < 1e-3
    return err


@pytest.mark.parametrize("name", ["tiny", "tinyjade"])
def test_small_configs(oracle, hip, name):
    hs, cfg = config_scene(name)
    p = B.params_from_config(cfg)
    _assert_parity(*_render_both(oracle, hip, hs, p))


def test_c1_cornell_full(oracle, hip):
    """configs[0]: Cornell box, 256x256, 64 spp - the full reference-runnable case."""
    hs, cfg = config_scene("C1")
    p = B.params_from_config(cfg)
    assert (p.width, p.height, p.spp) == (256, 256, 64)
    _assert_parity(*_render_both(oracle, hip, hs, p))


def test_c2_jade_statue_subset(oracle, hip):
    """configs[1] scene (70k-triangle jade statue) at a size the oracle finishes in seconds."""
    hs, cfg = config_scene("C2")
    p = B.params_from_config(cfg, spp=8)
    p.width = p.height = 128
    _assert_parity(*_render_both(oracle, hip, hs, p))


def test_nonsquare_and_partial_tiles(oracle, hip):
    """Width/height not multiples of 16 and W != H (the reference is square-only)."""
    hs, cfg = config_scene("tinyjade")
    p = B.params_from_config(cfg, spp=3)
    p.width, p.height = 45, 27
    _assert_parity(*_render_both(oracle, hip, hs, p))


def test_tile_partition_is_bit_exact(hip):
    """N-rank tile partition == 1-rank image, bit for bit (per-pixel RNG streams)."""
    hs, cfg = config_scene("tinyjade")
    p = B.params_from_config(cfg, spp=4)
    p.width, p.height = 70, 50
    with hip.scene(hs) as sc:
        full, full_b, st_full = sc.render(p)
        acc = np.zeros_like(full)
        acc_b = np.zeros_like(full_b)
        tot = {k: 0 for k in counters(st_full)}
        for r in range(3):
            q = B.params_from_config(cfg, spp=4, tile_rank=r, tile_nranks=3)
            q.width, q.height = 70, 50
            part, part_b, st = sc.render(q)
            assert not (np.asarray(acc != 0) & np.asarray(part != 0)).any()
            acc += part
            acc_b += part_b
            for k, v in counters(st).items():
                tot[k] += v
    assert np.array_equal(acc.view(np.uint32), full.view(np.uint32))
    assert np.array_equal(acc_b, full_b)
    assert tot == counters(st_full)


def test_render_multi_equals_single_device(hip):
    """jade_render_multi (one process, one scene per device; here every scene on the one GPU) is
    bit-identical to jade_render, whatever the number of shares."""
    hs, cfg = config_scene("tinyjade")
    p = B.params_from_config(cfg, spp=6)
    p.width, p.height = 70, 50
    with hip.scene(hs) as s0:
        full, full_b, st_full = s0.render(p)
        for ndev in (1, 2, 3):
            extra = [hip.scene(hs) for _ in range(ndev - 1)]
            try:
                rgb, bgr, st = B.render_multi(hip, [s0] + extra, p)
            finally:
                for e in extra:
                    e.close()
            assert np.array_equal(rgb.view(np.uint32), full.view(np.uint32)) and np.array_equal(bgr, full_b)
            assert counters(st) == counters(st_full)


def test_render_multi_rccl_path_on_one_device(hip, monkeypatch):
    """What of the RCCL gather can run on one GPU: librccl is found and loaded, ncclCommInitAll / group / destroy succeed
    for a single rank, the own share is placed, the frame equals jade_render.  (Two ranks: the next test.)"""
    monkeypatch.setenv("JADE_FORCE_RCCL", "1")
    hs, cfg = config_scene("tinyjade")
    p = B.params_from_config(cfg, spp=3)
    p.width, p.height = 40, 30
    with hip.scene(hs) as s0:
        full, full_b, st_full = s0.render(p)
        rgb, bgr, st = B.render_multi(hip, [s0], p)
    assert np.array_equal(rgb.view(np.uint32), full.view(np.uint32)) and np.array_equal(bgr, full_b)
    assert counters(st) == counters(st_full)


def test_render_multi_over_two_devices_rccl(hip):
    """jade_render_multi with scenes on DISTINCT devices: the shares meet through the RCCL gather (ncclCommInitAll +
    grouped send/recv).  Needs two GPUs; the round's GPU box has one, so this runs wherever more are visible."""
    if hip.device_count() < 2:
        pytest.skip("one GPU visible: the RCCL gather needs two distinct devices")
    hs, cfg = config_scene("tinyjade")
    p = B.params_from_config(cfg, spp=6)
    p.width, p.height = 70, 50
    with hip.scene(hs, device_id=0) as s0, hip.scene(hs, device_id=1) as s1:
        full, full_b, st_full = s0.render(p)
        rgb, bgr, st = B.render_multi(hip, [s0, s1], p)
    assert np.array_equal(rgb.view(np.uint32), full.view(np.uint32)) and np.array_equal(bgr, full_b)
    assert counters(st) == counters(st_full)


def test_progressive_equals_single_call(hip):
    """begin + N x step(spp) + resolve == one render of N*spp (RNG state and sums persist)."""
    hs, cfg = config_scene("tiny")
    p = B.params_from_config(cfg, spp=6)
    with hip.scene(hs) as sc:
        one, one_b, _ = sc.render(p)
        sc.begin(p)
        for _ in range(3):
            sc.step(2)
        prog, prog_b = sc.resolve()
    assert np.array_equal(one.view(np.uint32), prog.view(np.uint32))
    assert np.array_equal(one_b, prog_b)


def test_steps_may_carry_paths_over(hip):
    """A step may leave its last, longest paths to the next step (jade_render_flush, jade_rt.h): after
    flush the statistics of N steps equal those of one render of N*spp, and so does the image."""
    hs, cfg = config_scene("tinyjade")
    p = B.params_from_config(cfg, spp=12)
    p.width, p.height = 160, 120
    with hip.scene(hs) as sc:
        one, one_b, st_one = sc.render(p)
        sc.begin(p)
        st = _abi.Stats()
        for _ in range(3):
            sc.step(4, st)
        sc.flush(st)
        assert counters(st) == counters(st_one)
        sc.flush(st)  # nothing left: a second flush adds nothing
        assert counters(st) == counters(st_one)
        prog, prog_b = sc.resolve()
    assert np.array_equal(one.view(np.uint32), prog.view(np.uint32))
    assert np.array_equal(one_b, prog_b)


def test_result_independent_of_paths_in_flight(hip, monkeypatch):
    """The backend picks how many records per pixel work on a pixel's JADE_SAMPLE_LANES sample lanes
    (256 for a full 1080p frame on one GPU, 1024 for an eighth of it): the image must not depend on it."""
    hs, cfg = config_scene("tinyjade")
    p = B.params_from_config(cfg, spp=40)
    p.width, p.height = 24, 20
    ref = None
    for rpp in ("1", "8", "256", "1024"):
        monkeypatch.setenv("JADE_RECORDS_PER_PIXEL", rpp)
        with hip.scene(hs) as sc:
            rgb, bgr, st = sc.render(p)
        if ref is None:
            ref = (rgb, bgr, counters(st))
        else:
            assert np.array_equal(rgb.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(bgr, ref[1])
            assert counters(st) == ref[2]


def test_result_independent_of_shade_schedule(hip, monkeypatch):
    """A step's first pass is fused: light samples traced and shaded in one kernel, the rays of a wave walked as one packet
    (k_light_packet) or one lane per ray (k_light, JADE_LIGHT_PACKET=0), a packet given up after JADE_PACKET_BUDGET records
    handed to the wavefront passes; JADE_FUSED=0 runs the first pass as k_shade_lean + k_shade + k_trace passes instead, and
    JADE_SHADE_SPLIT=0 as k_shade alone over the active list (what later passes use anyway); list-mode passes run in
    batches with device-side counts or one host wait per pass (JADE_BATCH).  Every bit of the result must be the same."""
    for name, spp in (("tinyjade", 24), ("C1", 6)):
        hs, cfg = config_scene(name)
        p = B.params_from_config(cfg, spp=spp)
        p.width, p.height = 40, 36
        ref = None
        #            split fused batch packet budget wide (k_trace_wide for the walk=1 frame) tail (k_tail finishes short lists; 0: passes to the end)
        #            binned (k_shade deals its records by branch through LDS; 0: every thread runs its own record's whole bounce)
        #            records (k_trace refills from 48-B ray records the shading kernels wrote; 0: it gathers each ray through its queue entry)
        for v in (("1", "1", "1", "1", "32", "0", "1", "1", "1"), ("1", "1", "1", "0", "32", "1", "0", "1", "0"), ("1", "1", "1", "1", "3", "1", "1", "0", "1"), ("1", "1", "1", "1", "100000", "0", "0", "0", "0"),
                  ("1", "0", "1", "1", "32", "1", "0", "1", "1"), ("0", "1", "1", "1", "32", "0", "1", "1", "0"), ("1", "1", "0", "1", "32", "1", "0", "1", "1"), ("1", "1", "0", "1", "32", "0", "1", "0", "0"),
                  ("1", "0", "1", "1", "32", "0", "1", "0", "1"), ("0", "1", "0", "1", "32", "0", "0", "1", "1")):
            for key, val in zip(("JADE_SHADE_SPLIT", "JADE_FUSED", "JADE_BATCH", "JADE_LIGHT_PACKET", "JADE_PACKET_BUDGET", "JADE_WIDE", "JADE_TAIL", "JADE_SHADE_BINNED", "JADE_RAY_RECORDS"), v):
                monkeypatch.setenv(key, val)
            with hip.scene(hs) as sc:
                rgb, bgr, st = sc.render(p)
                early = sc.render(_with_walk(p, _abi.WALK_EARLY_EXIT))  # ... and so must the frame with early exits, in every schedule
                assert_cached_walk_equals_reference_walk(sc, p, (rgb, bgr, st))  # ... and with the occluder cache
            assert_early_exit_equals_reference_walk((rgb, bgr, st), early)
            if ref is None:
                ref = (rgb, bgr, counters(st))
            else:
                assert np.array_equal(rgb.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(bgr, ref[1]), v
                assert counters(st) == ref[2], v


def test_tail_kernel_finishes_what_passes_would(oracle, hip, monkeypatch):
    """k_tail (round 4): once the active list is short, ONE launch finishes its records - every wave shades and traces its own 64
    until they are out of samples - instead of a pass per bounce.  Same statements in the same order per record, so the frame,
    every counter and (reference walk) every node record and triangle test are those of the pass-by-pass schedule and of the
    oracle; progressive steps with carry-over and a flush included, and a threshold small enough that passes run first."""
    hs, cfg = config_scene("tinyjade")
    p = B.params_from_config(cfg, spp=40)
    p.width, p.height = 72, 56
    outs = {}
    for tail, tmax in (("0", "32768"), ("1", "32768"), ("1", "600"), ("1", "64")):
        monkeypatch.setenv("JADE_TAIL", tail)
        monkeypatch.setenv("JADE_TAIL_MAX", tmax)
        with hip.scene(hs) as sc:
            whole = sc.render(p)
            sc.begin(p)
            st = _abi.Stats()
            for spp in (16, 8, 16):
                sc.step(spp, st)
            sc.flush(st)
            steps = sc.resolve() + (st,)
            early = sc.render(_with_walk(p, _abi.WALK_EARLY_EXIT))
        assert_early_exit_equals_reference_walk(whole, early)
        assert np.array_equal(whole[0].view(np.uint32), steps[0].view(np.uint32)) and counters(whole[2]) == counters(steps[2])
        outs[(tail, tmax)] = whole
        if tail == "0":
            assert whole[2].tail_launches == 0 and whole[2].rays_tail == 0
        else:
            assert whole[2].tail_launches >= 1 and 0 < whole[2].rays_tail < whole[2].rays and whole[2].nodes_tail < whole[2].nodes_visited
    ref = outs[("0", "32768")]
    for k, o in outs.items():
        assert np.array_equal(o[0].view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(o[1], ref[1]), k
        assert counters(o[2]) == counters(ref[2]), k
    assert outs[("1", "64")][2].trace_launches > outs[("1", "32768")][2].trace_launches  # (a low threshold: passes first)
    with oracle.scene(hs) as so:
        r_o, b_o, st_o = so.render(p)
    assert counters(ref[2]) == counters(st_o) and rel_l2(ref[0], r_o) <= TOL


def test_result_independent_of_ray_ordering(hip, monkeypatch):
    """The ray queue may be ordered by (kind of ray, source triangle, octant) before k_trace takes it (the default for scenes
    whose tree does not fit the L2; JADE_SORT forces it either way): every result goes back to the ray's own slot, so image
    and counters must not move by a bit."""
    for name, spp in (("tinyjade", 16), ("C1", 4)):
        hs, cfg = config_scene(name)
        p = B.params_from_config(cfg, spp=spp)
        p.width, p.height = 64, 48
        ref = None
        for sort, keys_kernel in (("0", "0"), ("1", "0"), ("1", "1")):  # keys from the queueing kernel (the default) / from k_ray_keys
            monkeypatch.setenv("JADE_SORT", sort)
            monkeypatch.setenv("JADE_SORT_KEYS_KERNEL", keys_kernel)
            monkeypatch.setenv("JADE_SORT_MIN", "64")
            with hip.scene(hs) as sc:
                rgb, bgr, st = sc.render(p)
                early = sc.render(_with_walk(p, _abi.WALK_EARLY_EXIT))
                assert_cached_walk_equals_reference_walk(sc, p, (rgb, bgr, st))
            assert_early_exit_equals_reference_walk((rgb, bgr, st), early)
            if ref is None:
                ref = (rgb, bgr, counters(st))
            else:
                assert np.array_equal(rgb.view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(bgr, ref[1])
                assert counters(st) == ref[2]


def test_many_samples_per_lane_match_oracle(oracle, hip):
    """spp > JADE_SAMPLE_LANES: several samples per lane, summed in lane order by both backends."""
    hs, cfg = config_scene("tiny")
    p = B.params_from_config(cfg, spp=_abi.JADE_SAMPLE_LANES * 2 + 44)
    p.width, p.height = 12, 10
    _assert_parity(*_render_both(oracle, hip, hs, p))


def test_c5_deep_bvh_subset(oracle, hip):
    """configs[4] scene: 873,634-triangle "dragon" stand-in, BVH depth 25 (> the 24 LDS stack levels)."""
    hs, cfg = config_scene("C5")
    assert hs.n_triangles == 20 * 209 * 209 + 14 and hs.bvh_depth > 24
    p = B.params_from_config(cfg, spp=2)
    p.width, p.height = 96, 54
    _assert_parity(*_render_both(oracle, hip, hs, p))
    o, d, skip = _random_rays(hs, 20000, 77)
    with oracle.scene(hs) as so, hip.scene(hs) as sh:
        i_o, t_o, p_o, st_o = so.trace_rays(o, d, skip)
        i_h, t_h, p_h, st_h = sh.trace_rays(o, d, skip)
    assert np.array_equal(i_o, i_h) and (st_o.nodes_visited, st_o.tris_tested) == (st_h.nodes_visited, st_h.tris_tested)


def test_stack_spill_path(oracle):
    """A build with a 4-entry LDS stack: every traversal deeper than 4 goes through the global spill area."""
    import os
    from jaderaytracerendering_amd.backend import Backend, _LIBDIR
    be = Backend(os.path.join(_LIBDIR, "libjade_hip_stack4.so"))
    for name, size, spp in (("tinyjade", 32, 4), ("C2", 48, 2)):
        hs, cfg = config_scene(name)
        p = B.params_from_config(cfg, spp=spp)
        p.width = p.height = size
        _assert_parity(*_render_both(oracle, be, hs, p))


def test_preview_tone_operator(oracle, hip):
    """jade_render_resolve_ex with the GL preview's operator (pass3.fsh:8-18 == PathTrace.cu:669-672)."""
    hs, cfg = config_scene("tinyjade")
    p = B.params_from_config(cfg, spp=8)
    with oracle.scene(hs) as so, hip.scene(hs) as sh:
        outs = []
        for sc in (so, sh):
            sc.begin(p)
            sc.step(8)
            outs.append((sc.resolve(tonemap=_abi.TONEMAP_REINHARD, limit=1.5), sc.resolve()))
    (ro, rh) = outs
    assert np.abs(ro[0][1].astype(int) - rh[0][1].astype(int)).max() <= 1
    assert (ro[0][1] != ro[1][1]).mean() > 0.2          # a genuinely different curve from ACES
    assert np.array_equal(ro[0][0], ro[1][0])           # the linear mean is the same either way


def _random_rays(hs, n, seed):
    rng = np.random.default_rng(seed)
    v = hs.vertices().reshape(-1, 3)
    lo, hi = v.min(0), v.max(0)
    ctr, ext = (lo + hi) / 2, (hi - lo).max()
    o = (ctr + (rng.random((n, 3)) - 0.5) * ext * 1.5).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[: n // 4] *= rng.random((n // 4, 1)).astype(np.float32) * 5  # unnormalised, like shadow rays
    skip = rng.integers(-1, hs.n_triangles, n).astype(np.int32)
    return o, d, skip


@pytest.mark.parametrize("name", ["tiny", "C1", "C2"])
def test_trace_rays_bit_exact(oracle, hip, name):
    """hitBVH on raw rays: triangle index, hit point and distance bit-exact, V/T counters equal.  The distance is
    the device's own best `dist` - the value k_trace compared with `<` (hitArray, PathTrace.cu:787), exported by
    jade_trace_rays - not a host recomputation from the hit point; a miss keeps INF (PathTrace.cu:799) on both sides."""
    hs, _ = config_scene(name)
    o, d, skip = _random_rays(hs, 20000, 1234)
    # rays that start ON triangles and leave along axis directions (zero components -> inf slabs)
    v = hs.vertices()
    k = min(2000, hs.n_triangles)
    o[:k] = v[:k].mean(1)
    skip[:k] = np.arange(k)
    d[:k] = np.eye(3, dtype=np.float32)[np.arange(k) % 3] * np.where(np.arange(k) % 2, 1, -1)[:, None]
    with oracle.scene(hs) as so, hip.scene(hs) as sh:
        i_o, t_o, p_o, st_o = so.trace_rays(o, d, skip)
        i_h, t_h, p_h, st_h = sh.trace_rays(o, d, skip)
    assert np.array_equal(i_o, i_h)
    hitm = i_o >= 0
    assert hitm.sum() > 100
    assert np.array_equal(p_o[hitm].view(np.uint32), p_h[hitm].view(np.uint32))
    assert np.array_equal(t_o.view(np.uint32), t_h.view(np.uint32))          # misses included: INF
    assert (t_h[~hitm] == np.float32(2147483647.0)).all() and (~hitm).sum() > 100
    assert (st_o.nodes_visited, st_o.tris_tested) == (st_h.nodes_visited, st_h.tris_tested)


def test_trace_rays_equal_distances_keep_the_first_in_walk_order(oracle, hip):
    """Every triangle of this scene exists twice at the same place (two copies of one object), so nearly every hit has a
    twin at exactly the same distance, most of them in another leaf.  hitArray keeps the first one it meets (strict "<",
    PathTrace.cu:787; leaves in walk order, :806-856): k_trace tests the leaves of a ray on whatever lanes are free and in
    any order, and has to end up with the same triangle - the (distance, leaf sequence, place in the leaf) rule of
    resolve_hit."""
    from jaderaytracerendering_amd import host as H
    b = J.SceneBuilder()
    b.config("tiny")
    mat = H.material(brdf=(0.5,) * 3)
    for _ in range(2):
        b.add_proc("geodesic", 3, mat, H.transform_matrix(trans=(0.1, -1.2, 1.0), scale=(1.1, 1.1, 1.1)))
    b.set_env_sky(16, 8)
    hs = b.build()
    o, d, skip = _random_rays(hs, 60000, 77)
    skip[:] = -1
    with oracle.scene(hs) as so, hip.scene(hs) as sh:
        i_o, t_o, p_o, st_o = so.trace_rays(o, d, skip)
        i_h, t_h, p_h, st_h = sh.trace_rays(o, d, skip)
    hitm = i_o >= 0
    # the twins make ties common: count the hits whose distance some OTHER triangle reaches too is not possible from here,
    # but an index mismatch with equal distances is exactly what a wrong tie rule would produce
    assert hitm.sum() > 1000
    assert np.array_equal(t_o.view(np.uint32), t_h.view(np.uint32))
    assert np.array_equal(i_o, i_h)
    assert np.array_equal(p_o[hitm].view(np.uint32), p_h[hitm].view(np.uint32))
    assert (st_o.nodes_visited, st_o.tris_tested) == (st_h.nodes_visited, st_h.tris_tested)


def test_tree_with_missing_children(oracle, hip):
    """The reference's BVHNode_cu allows "child 0" = no child (PathTrace.cu:809-832 tests `left > 0` / `right > 0`); its own
    builder never produces one, so a tree is pruned by hand here: k_trace / k_light then take the general form of the node
    step for every wave (jade_scene_create sets DevScene.general_walk), which neither counts nor enters a missing child."""
    import copy
    hs, cfg = config_scene("tiny")
    hs = copy.deepcopy(hs)
    nd = hs.node_i32()  # left, right, n, index | aa | bb
    internal = [i for i in range(1, hs.n_nodes) if nd[i, 2] <= 0 and nd[i, 0] > 0 and nd[i, 1] > 0]
    assert len(internal) > 6
    nd[internal[2], 1] = 0   # a right child gone ...
    nd[internal[5], 0] = 0   # ... and a left one
    o, d, skip = _random_rays(hs, 30000, 5)
    with oracle.scene(hs) as so, hip.scene(hs) as sh:
        i_o, t_o, p_o, st_o = so.trace_rays(o, d, skip)
        i_h, t_h, p_h, st_h = sh.trace_rays(o, d, skip)
    assert np.array_equal(i_o, i_h) and np.array_equal(t_o.view(np.uint32), t_h.view(np.uint32))
    assert (st_o.nodes_visited, st_o.tris_tested) == (st_h.nodes_visited, st_h.tris_tested)
    p = B.params_from_config(cfg, spp=4)
    _assert_parity(*_render_both(oracle, hip, hs, p))


def test_refraction_material(oracle, hip):
    """DIR_REFRACT (refract_mode 2, PathTrace.cu:1180-1262): a glass ball in the Cornell box."""
    from jaderaytracerendering_amd import host as H
    b = J.SceneBuilder()
    cfg = b.config("tiny")
    glass = H.material(brdf=(0.05,) * 3, reflex_mode=1, refract_mode=2, refract_rate=(0.9, 0.95, 0.9),
                       refract_albedo=(0.3,) * 3, refract_index=1.5)
    b.add_proc("geodesic", 4, glass, H.transform_matrix(trans=(0.2, -1.6, 1.2), scale=(0.9, 0.9, 0.9)))
    b.set_env_sky(64, 32)
    hs = b.build()
    p = B.params_from_config(cfg, spp=8)
    p.width = p.height = 48
    _assert_parity(*_render_both(oracle, hip, hs, p))


def test_glass_statue_config_matches_oracle(oracle, hip):
    """Config C3G (bench.py's `glass_statue` extra): C3's 69,634-triangle scene with the statue made of DIR_REFRACT glass, the
    camera moved in so that the serial chain of internal reflections / refractions (PathTrace.cu:1202-1262) is most of the
    frame's work; the chain must be there (more refraction rays than samples) and everything equal to the oracle."""
    hs, cfg = config_scene("C3G")
    centre = hs.vertices()[hs.tri_i32()[:, 0] == 0].reshape(-1, 3).mean(0)
    eye = [float(x) for x in centre - 0.22 * (-np.array(cfg.camera[8:11], np.float32))]
    p = B.make_params(80, 48, 6, eye, list(cfg.camera))
    o, h = _render_both(oracle, hip, hs, p)
    _assert_parity(o, h)
    st = h[2]
    assert st.rays_refract > st.samples and st.rays_shadow == 0 and st.rays_env == 0


def test_full_size_properties(hip):
    """BASELINE full image size (1920x1080) on the C3 scene, 1 spp: size-independent properties.

    The oracle cannot finish this size in seconds, so check what must hold at any
    size: one primary ray per sample, counters consistent, image finite, and a
    re-render is bit-identical (determinism)."""
    hs, cfg = config_scene("C3")
    p = B.params_from_config(cfg, spp=1)
    assert (p.width, p.height) == (1920, 1080)
    with hip.scene(hs) as sc:
        a, ab, st = sc.render(p)
        b2, bb, st2 = sc.render(p)
    assert st.rays_primary == st.samples == 1920 * 1080
    assert st.shaded_hits >= 1 and st.rays_secondary >= st.shaded_hits // 2
    assert st.nodes_visited >= st.rays_primary + st.rays_secondary
    assert np.isfinite(a).all()  # may be negative: the reference's exit Fresnel is R0 - (1-R0)(..)^5, PathTrace.cu:1102
    assert np.array_equal(a.view(np.uint32), b2.view(np.uint32)) and np.array_equal(ab, bb)
    assert counters(st) == counters(st2)


def test_c5_full_size_properties(hip):
    """BASELINE configs[4] at its full frame: the 873,634-triangle scene at 3840x2160, 1 spp, on one GPU (102 GB of
    partial sums).  Same size-independent properties as above; the oracle checks this scene on a 96x54 subset."""
    hs, cfg = config_scene("C5")
    p = B.params_from_config(cfg, spp=1)
    assert (p.width, p.height) == (3840, 2160) and hs.n_triangles == 873634
    with hip.scene(hs) as sc:
        a, ab, st = sc.render(p)
        b2, bb, st2 = sc.render(p)
    assert st.rays_primary == st.samples == 3840 * 2160
    assert st.rays_secondary == st.rays_shadow + st.rays_env + st.rays_indirect + st.rays_mirror + st.rays_refract
    assert st.shaded_hits >= 1 and st.nodes_visited >= st.rays_primary + st.rays_secondary
    assert np.isfinite(a).all()
    assert np.array_equal(a.view(np.uint32), b2.view(np.uint32)) and np.array_equal(ab, bb)
    assert counters(st) == counters(st2)
