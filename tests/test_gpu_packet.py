"""The packet form of the walk (jade_trace.h, packet_trace: what the fused first pass k_light_packet runs) on raw rays.

A wave's 64 rays walk the tree together with one scalar cursor and stack; a ray's own order of leaves (which decides
hitArray's ties, PathTrace.cu:787, :816) is carried as a path key instead.  Bar, per RAY: triangle index, distance and hit
point bit-exact, and the ray's own counts of node records (V) and triangle tests (T) equal to the oracle's - a lane must
take part in exactly the nodes its ray enters, whatever the other 63 do."""
import ctypes as C

import numpy as np
import pytest

from conftest import J, config_scene
from jaderaytracerendering_amd import host as H

pytestmark = pytest.mark.gpu


def _packet_rays(hip, sc, o, d, skip):
    fn = hip.lib.jade_debug_packet_rays  # libjade_hip_debug.so only (not part of jade_rt.h)
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 9
    n = len(o)
    hit, dist, pt = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    v, t = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    hip.check(fn(sc._h, n, o.ctypes.data, d.ctypes.data, skip.ctypes.data, hit.ctypes.data, dist.ctypes.data, pt.ctypes.data,
                 v.ctypes.data, t.ctypes.data, None))
    return hit, dist, pt, v, t


def _oracle_per_ray(so, o, d, skip):
    n = len(o)
    hit, dist, pt = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    v, t = np.zeros(n, np.int64), np.zeros(n, np.int64)
    for i in range(n):
        ii, dd, pp, st = so.trace_rays(o[i:i + 1], d[i:i + 1], skip[i:i + 1])
        hit[i], dist[i], pt[i], v[i], t[i] = ii[0], dd[0], pp[0], st.nodes_visited, st.tris_tested
    return hit, dist, pt, v, t


def _rays(hs, packets, seed):
    """Packets of 64: even ones coherent like camera rays (one origin, a cone of directions), every third incoherent,
    some leaving triangles (skip), some with zero direction components (inf slabs)."""
    rng = np.random.default_rng(seed)
    v = hs.vertices()
    flat = v.reshape(-1, 3)
    lo, hi = flat.min(0), flat.max(0)
    n = 64 * packets
    o, d = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    skip = np.full(n, -1, np.int32)
    for p in range(packets):
        s = slice(64 * p, 64 * p + 64)
        oc = lo + (hi - lo) * (rng.random(3) * 1.4 - 0.2)
        dc = rng.normal(size=3)
        dc /= np.linalg.norm(dc)
        if p % 3 == 2:
            o[s] = lo + (hi - lo) * (rng.random((64, 3)) * 1.4 - 0.2)
            d[s] = rng.normal(size=(64, 3))
        else:
            o[s] = oc + rng.normal(size=(64, 3)) * (0.0 if p % 2 else 0.03) * (hi - lo).max()
            d[s] = dc + rng.normal(size=(64, 3)) * 0.15
        if p % 5 == 4:  # rays that start ON triangles and skip them, as mirror rays do
            k = rng.integers(0, hs.n_triangles, 64)
            o[s] = v[k].mean(1)
            skip[s] = k
        if p % 7 == 6:
            d[s][:, p % 3] = 0.0
    return o, d, skip


@pytest.mark.parametrize("name", ["tiny", "tinyjade", "C2"])
def test_packet_walk_matches_oracle_ray_by_ray(oracle, hip_debug, name):
    hs, _ = config_scene(name)
    o, d, skip = _rays(hs, 42 if name != "C2" else 24, 11)
    with oracle.scene(hs) as so, hip_debug.scene(hs) as sh:
        want = _oracle_per_ray(so, o, d, skip)
        got = _packet_rays(hip_debug, sh, o, d, skip)
    assert np.array_equal(got[0], want[0])
    hit = want[0] >= 0
    assert hit.sum() > 50 and (~hit).sum() > 50
    assert np.array_equal(got[1].view(np.uint32), want[1].view(np.uint32))              # distance, INF on a miss
    assert np.array_equal(got[2][hit].view(np.uint32), want[2][hit].view(np.uint32))    # hit point
    assert np.array_equal(got[3].astype(np.int64), want[3]), "node records per ray"
    assert np.array_equal(got[4].astype(np.int64), want[4]), "triangle tests per ray"


def test_packet_with_a_tie_between_leaves_is_given_up(oracle, hip_debug):
    """Every triangle of the ball twelve times at the same place, eight triangles to a leaf: every hit on it has twins at exactly
    the same distance in other leaves.  The packet meets leaves in ITS order (left first), a ray's own order is near-first, and hitArray gives a tie
    to the leaf the ray met first (strict "<", PathTrace.cu:787): a packet in which two leaves tie for some ray's best distance
    is given up - k_light_packet hands its rays to the wavefront passes, which walk them in the reference's order (the frames of
    such scenes: tests/test_gpu_early_exit.py, twin geometry).  Here: a packet is either given up as a whole (-3) or right in
    every lane."""
    b = J.SceneBuilder()
    b.config("tiny")
    mat = H.material(brdf=(0.5,) * 3)
    for _ in range(12):  # twelve copies of every triangle and eight triangles to a leaf: the copies MUST sit in several leaves
        b.add_proc("geodesic", 1, mat, H.transform_matrix(trans=(0.1, -1.2, 1.0), scale=(1.1, 1.1, 1.1)))
    b.set_env_sky(16, 8)
    hs = b.build()
    o, d, skip = _rays(hs, 60, 3)
    skip[:] = -1
    with oracle.scene(hs) as so, hip_debug.scene(hs) as sh:
        want = _oracle_per_ray(so, o, d, skip)
        got = _packet_rays(hip_debug, sh, o, d, skip)
    assert (want[0] >= 0).sum() > 500
    given_up = (got[0] == -3).reshape(-1, 64)
    assert (given_up.all(1) | ~given_up.any(1)).all(), "a packet is given up as a whole"
    gu = given_up.all(1)
    assert gu.sum() >= 5 and (~gu).sum() >= 5, (int(gu.sum()), int((~gu).sum()))
    keep = np.repeat(~gu, 64)
    assert np.array_equal(got[1][keep].view(np.uint32), want[1][keep].view(np.uint32))
    assert np.array_equal(got[0][keep], want[0][keep])
    assert np.array_equal(got[3][keep].astype(np.int64), want[3][keep]) and np.array_equal(got[4][keep].astype(np.int64), want[4][keep])
    # a packet none of whose rays meets the twins has nothing to tie: it must not be given up
    no_hit = (want[0] < 0).reshape(-1, 64).all(1)
    assert not (gu & no_hit).any()
