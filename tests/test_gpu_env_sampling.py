"""jade_render_params.env_sampling = JADE_ENV_IMPORTANCE (jade_rt.h, ABI 7; SURVEY 8f rank 3: "optional importance sampling (non-parity mode)").

NOT a parity mode: the environment-visibility ray's direction is drawn proportionally to the sky's luminance instead of uniformly over
the hemisphere (PathTrace.cu:968-979), so the samples are not the reference's.  What must hold instead:
  * it estimates the SAME integral: at a high sample count the frame agrees with the reference estimator's within their noise;
  * under a sky with a sun it has less noise at equal samples (that is its point);
  * it is as deterministic as every other mode (both walks: the same bits), the counters add up (an environment ray whose direction
    lies on the wrong side of the surface is not traced), the oracle refuses it, an unknown value is refused."""
import numpy as np
import pytest

from conftest import B, J, assert_early_exit_equals_reference_walk, config_scene
from jaderaytracerendering_amd import _abi

pytestmark = pytest.mark.gpu


def _sky_lit_ball():
    """A diffuse ball and a diffuse slab under the procedural sky (gradient + sun lobe): no emitter, all light is environment light."""
    from jaderaytracerendering_amd import host as H
    cfg = J.SceneBuilder().config("tiny")  # (the camera only: that builder also holds the closed Cornell box, under which no sky is seen)
    b = J.SceneBuilder()
    grey = H.material(brdf=(0.6, 0.55, 0.5))
    b.add_proc("geodesic", 3, grey, H.transform_matrix(trans=(0.1, -1.2, 1.0), scale=(1.1, 1.1, 1.1)))
    b.add_proc("box", 0, H.material(brdf=(0.5, 0.5, 0.5)), H.transform_matrix(trans=(0.0, -2.3, 1.0), scale=(9.0, 0.2, 9.0)))
    b.set_env_sky(64, 32)
    return b.build(), cfg


def _params(cfg, spp, env, walk=_abi.WALK_EARLY_EXIT, frame=0):
    p = B.params_from_config(cfg, spp=spp)
    p.width, p.height = 48, 40
    p.walk, p.env_sampling, p.frame = walk, env, frame
    return p


def test_importance_sampling_estimates_the_same_integral_with_less_noise(hip):
    hs, cfg = _sky_lit_ball()
    assert len(hs.a["emit"]) == 0
    with hip.scene(hs) as sc:
        ref, _, st_ref = sc.render(_params(cfg, 16384, _abi.ENV_REFERENCE))
        imp, _, st_imp = sc.render(_params(cfg, 16384, _abi.ENV_IMPORTANCE))
        lo_ref = [sc.render(_params(cfg, 32, _abi.ENV_REFERENCE, frame=1000 * k))[0] for k in range(1, 5)]
        lo_imp = [sc.render(_params(cfg, 32, _abi.ENV_IMPORTANCE, frame=1000 * k))[0] for k in range(1, 5)]
    lit = (ref != imp).any(-1)  # pixels that see a surface (a pixel that sees the sky draws no environment direction: the same bits in both)
    assert lit.sum() > 400, lit.sum()
    # the same integral: total energy and the image agree within the two estimators' noise at 16 k samples
    assert abs(imp[lit].sum() / ref[lit].sum() - 1) < 0.01, (imp[lit].sum(), ref[lit].sum())
    rel = np.sqrt(((imp[lit] - ref[lit]) ** 2).sum() / (ref[lit] ** 2).sum())
    assert rel < 0.03, rel
    # less noise at equal samples (32 spp, four independent frames each, against the mean of the two converged frames)
    truth = 0.5 * (ref + imp)
    mse_ref = np.mean([((x[lit] - truth[lit]) ** 2).mean() for x in lo_ref])
    mse_imp = np.mean([((x[lit] - truth[lit]) ** 2).mean() for x in lo_imp])
    print(f"mean squared error at 32 spp: uniform hemisphere {mse_ref:.4g}, by importance {mse_imp:.4g} ({mse_ref / mse_imp:.2f}x)")
    assert mse_imp < 0.8 * mse_ref
    # an environment ray on the wrong side of the surface is not traced: fewer environment rays, and the counts still add up
    assert 0 < st_imp.rays_env < st_ref.rays_env
    for st in (st_ref, st_imp):
        assert st.rays_secondary == st.rays_shadow + st.rays_env + st.rays_indirect + st.rays_mirror + st.rays_refract


@pytest.mark.parametrize("name", ["tinyjade", "C2"])
def test_importance_sampling_is_deterministic_and_walk_independent(hip, name):
    """Every branch that draws an environment direction (SSS-diffuse, BSSRDF, diffuse) with both walks and the occluder cache: the same
    bits; a second render: the same bits; and a frame that differs from the reference estimator's (it IS another estimator)."""
    hs, cfg = config_scene(name)
    p = B.params_from_config(cfg, spp=6)
    p.width, p.height = 64, 48
    p.env_sampling = _abi.ENV_IMPORTANCE
    with hip.scene(hs) as sc:
        a = sc.render(p)
        b = sc.render(p)
        q = type(p).from_buffer_copy(p)
        q.walk = _abi.WALK_EARLY_EXIT
        e = sc.render(q)
        q.walk = _abi.WALK_EARLY_EXIT_CACHED
        c1 = sc.render(q)
        c2 = sc.render(q)
        r = type(p).from_buffer_copy(p)
        r.env_sampling = _abi.ENV_REFERENCE
        plain = sc.render(r)
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)) and np.isfinite(a[0]).all()
    assert_early_exit_equals_reference_walk(a, e)
    assert_early_exit_equals_reference_walk(a, c1, fewer=False)
    assert_early_exit_equals_reference_walk(a, c2, fewer=False)
    assert not np.array_equal(a[0], plain[0])
    assert a[2].rays_env < plain[2].rays_env and a[2].rays_primary == plain[2].rays_primary


def test_oracle_refuses_importance_sampling_and_unknown_values_are_refused(oracle, hip):
    hs, cfg = config_scene("tiny")
    p = B.params_from_config(cfg, spp=1)
    p.env_sampling = _abi.ENV_IMPORTANCE
    with oracle.scene(hs) as so:
        with pytest.raises(B.JadeError, match="env_sampling"):
            so.render(p)
    p.env_sampling = 5
    with hip.scene(hs) as sc:
        with pytest.raises(B.JadeError):
            sc.render(p)
