"""What k_trace_wide rests on (jade_trace.h, "Wide walk"): hitAABB (PathTrace.cu:758-771) is monotone in float32.

A node's box contains its children's boxes exactly (the builder takes min / max of floats), every step of the slab test is a
monotone function of the box's coordinates, and the return rule keeps "value > 0" under enlargement - so a ray that meets a
node's box meets its parent's, with the values as computed.  Checked here with a numpy restatement of hitAABB in float32 (the
reference's ternaries, NaN going to the second operand) that is first compared with the oracle's own node counts, then applied
to every node of a scene for rays of every kind the integrator makes (normalised or not, zero components, origins on triangles).
Rays with a non-finite 1/d are left out of the claim: k_trace gives them the NaN-faithful binary unit."""
import numpy as np
import pytest

from conftest import config_scene


def _slab(aa, bb, o, d):
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = (np.float32(1.0) / d).astype(np.float32)
        f = ((bb - o) * inv).astype(np.float32)
        n = ((aa - o) * inv).astype(np.float32)
    tmax = np.where(f > n, f, n)
    tmin = np.where(f < n, f, n)
    lt = lambda a, b: np.where(a < b, a, b)  # noqa: E731  (std::min: a < b ? a : b)
    gt = lambda a, b: np.where(a > b, a, b)  # noqa: E731
    t1 = lt(tmax[:, 0], lt(tmax[:, 1], tmax[:, 2]))
    t0 = gt(tmin[:, 0], gt(tmin[:, 1], tmin[:, 2]))
    return np.where(t1 >= t0, np.where(t0 > 0, t0, t1), np.float32(-1))


@pytest.mark.parametrize("name", ["tinyjade", "C1"])
def test_a_ray_that_meets_a_box_meets_its_parents(oracle, name):
    hs, _ = config_scene(name)
    ni, nf = hs.node_i32(), hs.node_f32()
    left, right, cnt = ni[:, 0], ni[:, 1], ni[:, 2]
    aa, bb = nf[:, 4:7].astype(np.float32), nf[:, 7:10].astype(np.float32)
    N = len(ni)
    parent = np.zeros(N, np.int64)
    for i in range(1, N):
        if cnt[i] == 0:
            for c in (left[i], right[i]):
                if c > 0:
                    parent[c] = i
    kids = np.nonzero(parent > 0)[0]
    assert (aa[parent[kids]] <= aa[kids]).all() and (bb[parent[kids]] >= bb[kids]).all()  # containment is exact
    rng = np.random.default_rng(5)
    v = hs.vertices()
    flat = v.reshape(-1, 3)
    lo, hi = flat.min(0), flat.max(0)
    n_rays = 600
    o = (lo + (hi - lo) * (rng.random((n_rays, 3)) * 1.6 - 0.3)).astype(np.float32)
    o[::3] = v[rng.integers(0, hs.n_triangles, len(o[::3]))].mean(1)
    d = rng.normal(size=(n_rays, 3)).astype(np.float32)
    d[::5] *= rng.random((len(d[::5]), 1)).astype(np.float32) * 7  # not normalised, like shadow rays
    skip = np.full(n_rays, -1, np.int32)
    met_total = 0
    with oracle.scene(hs) as so:
        for r in range(n_rays):
            met = _slab(aa, bb, o[r], d[r]) > 0
            met[1] = True  # the root is entered without a test
            assert not (met[kids] & ~met[parent[kids]]).any(), f"ray {r}: a box is met whose parent's is not"
            # the restatement agrees with the oracle: V = 1 + 2 x (internal nodes entered) in a tree without missing children
            entered = met & (cnt == 0)
            entered[0] = False
            _, _, _, st = so.trace_rays(o[r:r + 1], d[r:r + 1], skip[r:r + 1])
            assert st.nodes_visited == 1 + 2 * int(entered[1:].sum() if cnt[1] == 0 else 0) or cnt[1] > 0
            met_total += int(met[kids].sum())
    assert met_total > 1000


def test_met_rule_equals_the_returned_value_being_positive():
    """jade_trace.h asks of hitAABB's value r = (t1 >= t0) ? ((t0 > 0) ? t0 : t1) : -1 (PathTrace.cu:769-770) only "r > 0", as
    `t1 >= t0 && t1 > 0` (slab_met), and forms r = (t0 > 0) ? t0 : t1 (slab_dist) only where two boxes are both met.  Both
    restatements are checked here over every pair of a set of float32 values that holds the special ones (signed zeros, infinities,
    NaN, denormals, neighbours of 0 and 1) and random ones of every magnitude."""
    rng = np.random.default_rng(11)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.17549435e-38, -1.17549435e-38, 1.0, -1.0,
                        np.nextafter(np.float32(1), np.float32(2)), np.nextafter(np.float32(1), np.float32(0)), 3.4028235e38, -3.4028235e38], np.float32)
    rnd = (rng.normal(size=400) * np.exp(rng.uniform(-80, 80, size=400))).astype(np.float32)
    vals = np.concatenate([special, rnd, -rnd[:50]])
    t0, t1 = np.meshgrid(vals, vals, indexing="ij")
    with np.errstate(invalid="ignore"):
        r = np.where(t1 >= t0, np.where(t0 > 0, t0, t1), np.float32(-1))
        met = (t1 >= t0) & (t1 > 0)
        assert np.array_equal(r > 0, met)
        dist = np.where(t0 > 0, t0, t1)
        assert np.array_equal(r[met].view(np.uint32), dist[met].view(np.uint32))  # where a box is met, slab_dist IS the value, bit for bit
