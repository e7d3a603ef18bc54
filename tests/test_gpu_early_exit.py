"""jade_render_params.walk = JADE_WALK_EARLY_EXIT (jade_rt.h): shadow and environment-visibility queries end at the first
recorded hit that settles what the integrator asks of them (PathTrace.cu:957 / 981 and their twins in the other branches).

The claim is exactness, not a tolerance: the frame, every ray count and every sample are those of the reference's walk; only
the node records read and the triangle tests made are fewer.  Checked here piece by piece -
  * the limit k_shade hands to k_trace for a shadow ray IS the distance k_trace / the oracle compute for the emitter (bits);
  * k_trace with a limit per ray: the reference's answer wherever the nearest hit is not nearer than the limit, otherwise a
    recorded hit nearer than the limit;
  * whole frames: tests/test_gpu_parity.py renders every one of its scenes with both walks (and every schedule with both),
    tests/test_gpu_bench_schedule.py the benchmarked frames; here the frames where early exits matter most (the statue
    filling the frame) and the oracle beside them."""
import ctypes as C

import numpy as np
import pytest

from conftest import B, J, assert_cached_walk_equals_reference_walk, assert_early_exit_equals_reference_walk, config_scene, counters, rel_l2, WALK_KEYS
from jaderaytracerendering_amd import _abi

pytestmark = pytest.mark.gpu
INF = np.float32(2147483647.0)


def _shadow_limit(hip, sc, o, d, tri):
    fn = hip.lib.jade_debug_shadow_limit  # libjade_hip_debug.so only (not part of jade_rt.h)
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 4
    out = np.zeros(len(o), np.float32)
    hip.check(fn(sc._h, len(o), o.ctypes.data, d.ctypes.data, tri.ctypes.data, out.ctypes.data))
    return out


def _trace_limit(hip, sc, o, d, skip, limit, cached=False):
    fn = hip.lib.jade_debug_trace_rays_cached if cached else hip.lib.jade_debug_trace_rays_limit  # libjade_hip_debug.so only (not part of jade_rt.h)
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 8
    n = len(o)
    hit, dist, pt = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    st = _abi.Stats()
    hip.check(fn(sc._h, n, o.ctypes.data, d.ctypes.data, skip.ctypes.data, limit.ctypes.data, hit.ctypes.data, dist.ctypes.data,
                 pt.ctypes.data, C.byref(st)))
    return hit, dist, pt, st


def _aimed_rays(hs, n, seed):
    """Rays as the jade branches aim them at an emitter: from a point (on a triangle it then skips, or in space) to a random
    point of a target triangle, direction NOT normalised (PathTrace.cu:955)."""
    rng = np.random.default_rng(seed)
    v = hs.vertices()
    flat = v.reshape(-1, 3)
    lo, hi = flat.min(0), flat.max(0)
    tri = rng.integers(0, hs.n_triangles, n).astype(np.int32)
    r = rng.random((n, 2)).astype(np.float32)
    flip = r.sum(1) > 1
    r[flip] = 1 - r[flip]
    target = v[tri, 0] + (v[tri, 1] - v[tri, 0]) * r[:, :1] + (v[tri, 2] - v[tri, 0]) * r[:, 1:]
    o = (lo + (hi - lo) * (rng.random((n, 3)) * 1.6 - 0.3)).astype(np.float32)
    skip = np.full(n, -1, np.int32)
    k = n // 3  # a third leave a triangle
    src = rng.integers(0, hs.n_triangles, k)
    o[:k] = v[src].mean(1)
    skip[:k] = src
    d = (target - o).astype(np.float32)
    d[::7, 1] = 0.0  # some with a zero component (an infinite slab, and a target missed)
    return np.ascontiguousarray(o), np.ascontiguousarray(d), skip, tri


@pytest.mark.parametrize("name", ["tinyjade", "C1", "C2"])
def test_shadow_limit_is_the_walks_own_distance(oracle, hip_debug, name):
    """Wherever the walk's nearest hit IS the target, its distance and the limit are the same float; a limit is never a value
    the walk could not have recorded (0 < limit <= INF); and a target the ray misses gives INF."""
    hip = hip_debug  # (the same kernels + the jade_debug_* entry points)
    hs, _ = config_scene(name)
    o, d, skip, tri = _aimed_rays(hs, 30000, 5)
    with oracle.scene(hs) as so, hip.scene(hs) as sh:
        i_o, t_o, _, _ = so.trace_rays(o, d, skip)
        i_h, t_h, _, _ = sh.trace_rays(o, d, skip)
        lim = _shadow_limit(hip, sh, o, d, tri)
    assert np.array_equal(i_o, i_h) and np.array_equal(t_o.view(np.uint32), t_h.view(np.uint32))
    seen = i_o == tri
    assert seen.sum() > 2000
    assert np.array_equal(lim[seen].view(np.uint32), t_o[seen].view(np.uint32))
    assert (lim > 0).all() and (lim <= INF).all() and not np.isnan(lim).any()
    assert (lim == INF).sum() > 100  # rays that miss their target (the zeroed component)
    # a nearest hit nearer than the limit is never the target (the limit is the target's own distance)
    assert not (seen & (t_o < lim)).any()


@pytest.mark.parametrize("wide", ["0", "1"])
@pytest.mark.parametrize("name", ["tinyjade", "C2"])
def test_walk_with_a_limit_per_ray(oracle, hip_debug, name, wide, monkeypatch):
    """k_trace on raw rays, each with a limit: NaN (never: the reference's walk), INF (any recorded hit), a distance - with
    binary units and with wide ones (JADE_WIDE: four grandchildren per visit; by default only for trees outside the L2)."""
    hip = hip_debug
    monkeypatch.setenv("JADE_WIDE", wide)
    hs, _ = config_scene(name)
    n = 40000
    o, d, skip, _ = _aimed_rays(hs, n, 9)
    rng = np.random.default_rng(3)
    d[1::4] = rng.normal(size=(len(d[1::4]), 3)).astype(np.float32)  # a quarter aimed at nothing: misses
    with hip.scene(hs) as sh:
        i0, t0, p0, st0 = sh.trace_rays(o, d, skip)
        hitm = i0 >= 0
        assert hitm.sum() > 5000 and (~hitm).sum() > 2000
        limit = np.full(n, np.float32(np.nan))
        limit.view(np.int32)[:] = -1  # the marker of a ray whose nearest hit is wanted
        kind = rng.integers(0, 4, n)
        limit[kind == 1] = INF
        near = np.where(hitm, t0, 1.0).astype(np.float32)
        limit[kind == 2] = (near * rng.uniform(0.5, 1.0, n).astype(np.float32))[kind == 2]  # at or below the nearest hit: never reached
        limit[kind == 3] = (near * rng.uniform(1.0, 3.0, n).astype(np.float32))[kind == 3]  # at or beyond it
        i1, t1, p1, st1 = _trace_limit(hip, sh, o, d, skip, limit)
    ends = hitm & (t0 < limit)  # (False for a NaN limit)
    assert ends.sum() > 5000 and (~ends & hitm).sum() > 3000
    same = ~ends
    assert np.array_equal(i1[same], i0[same]) and np.array_equal(t1[same].view(np.uint32), t0[same].view(np.uint32))
    assert np.array_equal(p1[same & hitm].view(np.uint32), p0[same & hitm].view(np.uint32))
    assert (i1[ends] >= 0).all() and (t1[ends] < limit[ends]).all() and (t1[ends] >= t0[ends]).all()
    assert st1.nodes_visited < st0.nodes_visited and st1.tris_tested < st0.tris_tested


def _closeup(hs, cfg):
    centre = hs.vertices()[hs.tri_i32()[:, 0] == 0].reshape(-1, 3).mean(0)
    return [float(x) for x in centre - 0.22 * (-np.array(cfg.camera[8:11], np.float32))]


@pytest.mark.parametrize("wide", ["0", "1"])
def test_statue_closeup_frames_are_the_same_bits(oracle, hip, wide, monkeypatch):
    """The frame that is all jade paths (bench.py's statue_closeup view): both walks on the GPU bit for bit, the oracle beside
    them - and the early exits do leave out a good part of the work there."""
    monkeypatch.setenv("JADE_WIDE", wide)
    hs, cfg = config_scene("C3")
    eye = _closeup(hs, cfg)
    p = B.make_params(96, 64, 24, eye, list(cfg.camera))
    q = B.make_params(96, 64, 24, eye, list(cfg.camera), walk=_abi.WALK_EARLY_EXIT)
    with hip.scene(hs) as sh, oracle.scene(hs) as so:
        ref = sh.render(p)
        early = sh.render(q)
        cached = assert_cached_walk_equals_reference_walk(sh, p, ref)  # ... and with the occluder cache, cold then warm
        r_o, b_o, st_o = so.render(q)  # (the oracle ignores .walk)
    assert_early_exit_equals_reference_walk(ref, early)
    assert cached[2].rays_cached > 0.2 * (cached[2].rays_shadow + cached[2].rays_env)  # (warm: the second render of the pair)
    assert cached[2].nodes_visited < early[2].nodes_visited
    assert counters(ref[2]) == counters(st_o)
    assert rel_l2(early[0], r_o) <= 1e-4
    assert early[2].nodes_visited < 0.92 * ref[2].nodes_visited and early[2].tris_tested < 0.88 * ref[2].tris_tested
    assert {k: v for k, v in counters(early[2]).items() if k not in WALK_KEYS} == {k: v for k, v in counters(st_o).items() if k not in WALK_KEYS}


def test_progressive_steps_with_early_exits(hip):
    """Steps, carry-over and flush with early exits: the frame of one call, bit for bit (paths carried from one step into the
    next keep their queued rays' limits in the slots)."""
    hs, cfg = config_scene("tinyjade")
    p = B.params_from_config(cfg, spp=48)
    p.width, p.height = 40, 28
    q = type(p).from_buffer_copy(p)
    q.walk = _abi.WALK_EARLY_EXIT
    with hip.scene(hs) as sc:
        ref = sc.render(p)
        sc.begin(q)
        st = _abi.Stats()
        for spp in (16, 8, 24):
            sc.step(spp, st)
        sc.flush(st)
        rgb, bgr = sc.resolve()
        q.walk = _abi.WALK_EARLY_EXIT_CACHED  # ... and the same steps with the occluder cache
        sc.begin(q)
        st_c = _abi.Stats()
        for spp in (16, 8, 24):
            sc.step(spp, st_c)
        sc.flush(st_c)
        rgb_c, bgr_c = sc.resolve()
    assert_early_exit_equals_reference_walk(ref, (rgb, bgr, st))
    assert_early_exit_equals_reference_walk(ref, (rgb_c, bgr_c, st_c), fewer=False)


def test_unknown_walk_is_refused(hip):
    hs, cfg = config_scene("tiny")
    p = B.params_from_config(cfg, spp=1)
    p.walk = 7
    with hip.scene(hs) as sc:
        with pytest.raises(B.JadeError):
            sc.render(p)


def _twin_scene(twin_light=False):
    """Every triangle of the ball exists twelve times at the same place (and, with twin_light, a light too) while a leaf holds
    eight triangles: every hit on them has twins at exactly the same distance in OTHER leaves."""
    from jaderaytracerendering_amd import host as H
    b = J.SceneBuilder()
    cfg = b.config("tinyjade" if twin_light else "tiny")
    mat = H.material(brdf=(0.5,) * 3)
    for _ in range(12):
        b.add_proc("geodesic", 1, mat, H.transform_matrix(trans=(0.1, -1.2, 1.0), scale=(1.1, 1.1, 1.1)))
    if twin_light:
        light = H.material(emissive=(30.0,) * 3, brdf=(0.3,) * 3)
        for _ in range(12):
            b.add_proc("geodesic", 0, light, H.transform_matrix(trans=(0.4, 1.0, 1.3), scale=(0.3, 0.3, 0.3)))
    b.set_env_sky(16, 8)
    return b.build(), cfg


def test_wide_walk_ties_are_walked_again_in_the_references_order(oracle, hip_debug, monkeypatch):
    """With early exits k_trace walks four grandchildren per visit (jade_trace.h, "Wide walk") - an order of leaves that is not
    the reference's, which only hitArray's tie rule can see (strict "<": of two equal distances the one met first wins,
    PathTrace.cu:787).  A ray for whose best distance two leaves tie is walked again with binary units; on twin geometry that is
    nearly every ray, and index, distance and hit point must be the oracle's for every one of them."""
    hip = hip_debug
    monkeypatch.setenv("JADE_WIDE", "1")  # (by default only trees that do not fit the L2 get wide records)
    hs, _ = _twin_scene()
    rng = np.random.default_rng(77)
    flat = hs.vertices().reshape(-1, 3)
    lo, hi = flat.min(0), flat.max(0)
    n = 60000
    o = ((lo + hi) / 2 + (rng.random((n, 3)) - 0.5) * (hi - lo).max() * 1.5).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    skip = np.full(n, -1, np.int32)
    never = np.full(n, np.float32(np.nan))
    never.view(np.int32)[:] = -1  # the nearest hit is wanted: the reference's answer, whatever the form of the walk
    with oracle.scene(hs) as so, hip.scene(hs) as sh:
        i_o, t_o, p_o, _ = so.trace_rays(o, d, skip)
        i_w, t_w, p_w, _ = _trace_limit(hip, sh, o, d, skip, never)
    hitm = i_o >= 0
    assert hitm.sum() > 1000
    assert np.array_equal(t_o.view(np.uint32), t_w.view(np.uint32))
    assert np.array_equal(i_o, i_w)
    assert np.array_equal(p_o[hitm].view(np.uint32), p_w[hitm].view(np.uint32))


@pytest.mark.parametrize("wide", ["0", "1"])
@pytest.mark.parametrize("twin_light", [False, True])
def test_twin_geometry_frames_are_the_same_bits(oracle, hip, twin_light, wide, monkeypatch):
    """... and whole frames of such scenes with both walks: twin occluders, and twin EMITTERS - a shadow ray's limit is then
    reached exactly by the twin of the emitter it aims at, and which of the two the reference meets first decides whether the
    light is seen."""
    monkeypatch.setenv("JADE_WIDE", wide)
    hs, cfg = _twin_scene(twin_light)
    p = B.params_from_config(cfg, spp=12)
    p.width, p.height = 48, 40
    q = type(p).from_buffer_copy(p)
    q.walk = _abi.WALK_EARLY_EXIT
    with hip.scene(hs) as sh, oracle.scene(hs) as so:
        ref = sh.render(p)
        early = sh.render(q)
        assert_cached_walk_equals_reference_walk(sh, p, ref, renders=3)
        r_o, b_o, st_o = so.render(p)
    assert counters(ref[2]) == counters(st_o) and rel_l2(ref[0], r_o) <= 1e-4
    assert_early_exit_equals_reference_walk(ref, early, fewer=False)  # (a ray with a tie is walked twice by the wide form)


@pytest.mark.parametrize("wide", ["0", "1"])
@pytest.mark.parametrize("name", ["tinyjade", "C2"])
def test_cached_walk_answers_every_query_as_the_whole_walk_does(oracle, hip_debug, name, wide, monkeypatch):
    """The occluder cache on raw rays (jade_trace.h, "Occluder cache"): each ray leaves a source triangle with a limit - NaN: the
    nearest hit is wanted and the cache must not touch it; INF or a distance: a yes/no query, keyed by its source triangle.  The
    same batch is traced five times on one scene handle, so that later rounds start from the subtrees earlier rounds cached.  What
    must hold every time: a query says "yes" (a recorded hit below the limit) exactly when the reference's nearest hit is below the
    limit, and then reports a hit below the limit that is not nearer than the reference's; otherwise index, distance and hit
    point are the reference's bits."""
    monkeypatch.setenv("JADE_WIDE", wide)
    hip = hip_debug
    hs, _ = config_scene(name)
    n = 40000
    o, d, skip, _ = _aimed_rays(hs, n, 21)
    rng = np.random.default_rng(8)
    # every ray leaves a triangle (the key), few distinct sources so that keys repeat
    v = hs.vertices()
    src = rng.integers(0, min(hs.n_triangles, 600), n).astype(np.int32)
    o = np.ascontiguousarray(v[src].mean(1).astype(np.float32))
    d[1::3] = rng.normal(size=(len(d[1::3]), 3)).astype(np.float32)
    d = np.ascontiguousarray(d)
    skip = src
    with hip.scene(hs) as sh:
        flags = hip.lib.jade_debug_scene_flags(sh._h)
        assert flags & 1 and flags & 4, "the scene should have nested boxes and an occluder cache"
        i0, t0, p0, st0 = sh.trace_rays(o, d, skip)
        hitm = i0 >= 0
        assert hitm.sum() > 5000
        limit = np.full(n, np.float32(np.nan))
        limit.view(np.int32)[:] = -1
        kind = rng.integers(0, 4, n)
        limit[kind == 1] = INF
        near = np.where(hitm, t0, 1.0).astype(np.float32)
        limit[kind == 2] = (near * rng.uniform(0.5, 1.0, n).astype(np.float32))[kind == 2]
        limit[kind == 3] = (near * rng.uniform(1.0, 3.0, n).astype(np.float32))[kind == 3]
        ends = hitm & (t0 < limit)
        same = ~ends
        answered = []
        for rnd in range(5):
            i1, t1, p1, st1 = _trace_limit(hip, sh, o, d, skip, limit, cached=True)
            assert np.array_equal(i1[same], i0[same]) and np.array_equal(t1[same].view(np.uint32), t0[same].view(np.uint32)), rnd
            assert np.array_equal(p1[same & hitm].view(np.uint32), p0[same & hitm].view(np.uint32)), rnd
            assert (i1[ends] >= 0).all() and (t1[ends] < limit[ends]).all() and (t1[ends] >= t0[ends]).all(), rnd
            assert (i1[ends] != skip[ends]).all()
            answered.append(int(st1.rays_cached))
    assert ends.sum() > 5000
    assert answered[0] <= answered[-1] and answered[-1] > 0.3 * ends.sum(), answered  # the cache learns: warm rounds answer more
    assert answered[-1] <= ends.sum()  # ... and only ever answers "yes" queries


def _with_inflated_child_boxes(hs, every, pad):
    """A copy of the scene whose BVH has every `every`-th non-root node's box padded by `pad` on all sides: still a legal tree for
    hitBVH (a box only has to contain its triangles), but children now stick out of their parents."""
    from jaderaytracerendering_amd.host import HostScene
    arrays = {k: np.array(v, copy=True) for k, v in hs.a.items()}
    f = arrays["nodes"].view(np.float32)
    picked = np.arange(2, f.shape[0])[::every]
    f[picked, 4:7] -= np.float32(pad)
    f[picked, 7:10] += np.float32(pad)
    return HostScene(arrays, hs.bvh_depth, 0.0)


def test_tree_with_loose_boxes_is_walked_from_the_root_with_binary_units(oracle, hip_debug, monkeypatch):
    """jade_scene_create takes the caller's BVH.  A legal tree whose child boxes are NOT inside their parents' (padded, refitted) has
    leaves the reference prunes although their own box is met: neither the wide walk nor the occluder cache may be used on it
    (ADVICE r3).  Inflate one child box per level of C1's tree: the scene reports "not nested", gets neither wide records nor a
    cache, and every walk mode still gives the oracle's frame on the same loose tree."""
    monkeypatch.setenv("JADE_WIDE", "1")
    hs, cfg = config_scene("C1")
    loose = _with_inflated_child_boxes(hs, every=3, pad=0.35)
    p = B.params_from_config(cfg, spp=4)
    p.width, p.height = 48, 48
    with hip_debug.scene(loose) as sh, oracle.scene(loose) as so:
        assert hip_debug.lib.jade_debug_scene_flags(sh._h) == 0
        ref = sh.render(p)
        q = type(p).from_buffer_copy(p)
        q.walk = _abi.WALK_EARLY_EXIT
        assert_early_exit_equals_reference_walk(ref, sh.render(q))
        assert_cached_walk_equals_reference_walk(sh, p, ref)
        r_o, b_o, st_o = so.render(p)
    assert counters(ref[2]) == counters(st_o) and rel_l2(ref[0], r_o) <= 1e-4
    with hip_debug.scene(hs) as sh:
        assert hip_debug.lib.jade_debug_scene_flags(sh._h) == 7  # the tree as built: nested, wide records (JADE_WIDE=1), cache
