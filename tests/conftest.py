"""Shared fixtures.

Roles, fixed by the task's oracle rule:
  * `oracle`  - oracle/libjade_oracle.so, the CPU restatement of PathTrace.cu:669-1474.
                It is the CHECKER; nothing under jaderaytracerendering_amd/ imports it.
  * `hip`     - jaderaytracerendering_amd/lib/libjade_hip.so, the product.  GPU tests
                call it through the jade_rt.h C ABI and fail (never skip) when it is absent.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import jaderaytracerendering_amd as J  # noqa: E402
from jaderaytracerendering_amd import backend as B  # noqa: E402

ORACLE_LIB = os.path.join(ROOT, "oracle", "libjade_oracle.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """CPU-side libraries are cheap to (re)build; the HIP library is built by __graft_entry__.build()."""
    if not os.path.exists(ORACLE_LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    if not os.path.exists(os.path.join(ROOT, "jaderaytracerendering_amd", "lib", "libjade_host.so")):
        subprocess.check_call(["make", "-C", ROOT, "host"])


@pytest.fixture(scope="session")
def oracle():
    _ensure_built()
    return B.Backend(ORACLE_LIB)


@pytest.fixture(scope="session")
def hip():
    _ensure_built()
    be = B.hip()  # raises if libjade_hip.so is missing: no silent fallback
    assert be.name == "hip-gfx950"
    assert be.device_count() >= 1, "no HIP device visible"
    return be


@pytest.fixture(scope="session")
def hip_debug(hip):
    """libjade_hip_debug.so: the product's sources built with -DJADE_DEBUG_EXPORTS=1 (make hipvariants) - the same kernels plus the
    jade_debug_* entry points the piece-by-piece tests call.  libjade_hip.so itself exports none of them (tests/test_abi.py)."""
    path = os.path.join(ROOT, "jaderaytracerendering_amd", "lib", "libjade_hip_debug.so")
    assert os.path.exists(path), "libjade_hip_debug.so missing: run `make hipvariants` (or __graft_entry__.build())"
    return B.Backend(path)


_scene_cache = {}


def config_scene(name):
    _ensure_built()
    if name not in _scene_cache:
        _scene_cache[name] = J.build_config(name)
    return _scene_cache[name]


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-300))


COUNTER_KEYS = ("rays_primary", "rays_secondary", "nodes_visited", "tris_tested", "shaded_hits", "samples",
                "rays_shadow", "rays_env", "rays_indirect", "rays_mirror", "rays_refract")


def counters(st):
    return {k: getattr(st, k) for k in COUNTER_KEYS}


WALK_KEYS = ("nodes_visited", "tris_tested")  # what jade_render_params.walk may change (jade_rt.h, JADE_WALK_*)


def assert_early_exit_equals_reference_walk(ref, early, fewer=True):
    """(rgb, bgr, stats) of one frame rendered with JADE_WALK_REFERENCE and with JADE_WALK_EARLY_EXIT: every float of the
    radiance and every byte the same, every ray / sample / vertex count the same; node records and triangle tests fewer
    (fewer=False: a scene made of ties, whose rays the wide walk has to walk a second time)."""
    (r0, b0, s0), (r1, b1, s1) = ref, early
    assert np.array_equal(r0.view(np.uint32), r1.view(np.uint32)), "radiance differs between the two walks"
    assert np.array_equal(b0, b1)
    c0, c1 = counters(s0), counters(s1)
    assert {k: v for k, v in c0.items() if k not in WALK_KEYS} == {k: v for k, v in c1.items() if k not in WALK_KEYS}
    for k in WALK_KEYS:
        assert not fewer or c1[k] <= c0[k], k


def assert_cached_walk_equals_reference_walk(sc, params, ref, renders=2):
    """JADE_WALK_EARLY_EXIT_CACHED (jade_rt.h, ABI 7): yes/no queries first walk the subtrees in which earlier such queries found
    their answer.  Rendered `renders` times on the same scene handle - the first with a cold cache (or whatever an earlier render of
    this scene left in it), the next ones with what the first one learnt - every frame must be the reference walk's bit for bit,
    with every ray / sample / vertex count; node records and triangle tests may be anything (they count what was read, and a failed
    attempt is read twice).  Returns the last render."""
    from jaderaytracerendering_amd import _abi
    q = type(params).from_buffer_copy(params)
    q.walk = _abi.WALK_EARLY_EXIT_CACHED
    out = None
    for _ in range(renders):
        out = sc.render(q)
        assert_early_exit_equals_reference_walk(ref, out, fewer=False)
        assert out[2].rays_cached <= out[2].rays_shadow + out[2].rays_env
    return out


@pytest.fixture(scope="session")
def fpm():
    """ctypes handle on tests/native/fpmath_export.c, built with the mandatory flags."""
    import ctypes
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libfpmath_test.so")
    src = os.path.join(ROOT, "tests", "native", "fpmath_export.c")
    hdrs = [os.path.join(ROOT, "include", h) for h in ("jade_fpmath.h", "jade_rt.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", "-fno-fast-math",
                               "-I", os.path.join(ROOT, "include"), "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.t_dot.restype = ctypes.c_float
    lib.t_mixed.restype = ctypes.c_float
    lib.t_fmin.restype = ctypes.c_float
    lib.t_fmax.restype = ctypes.c_float
    lib.t_fmin.argtypes = lib.t_fmax.argtypes = [ctypes.c_float, ctypes.c_float]
    lib.t_selftest.argtypes = [ctypes.c_float]
    lib.t_seed.restype = ctypes.c_uint32
    lib.t_seed.argtypes = [ctypes.c_uint32] * 3
    return lib


def object_tiles(hs, cfg, width, height, obj=0):
    """Tile ids the vertices of object `obj` project into (host.object_tiles)."""
    from jaderaytracerendering_amd import host as H
    return H.object_tiles(hs, cfg.eye, cfg.camera, width, height, obj)


def oracle_tile_filter(oracle_scene, tile_ids):
    """oracle/jade_oracle.c, jade_oracle_set_tile_filter: a checker-only export (the product has no such entry point)."""
    import ctypes
    fn = oracle_scene.backend.lib.jade_oracle_set_tile_filter
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
    ids = np.ascontiguousarray(tile_ids, np.int32)
    oracle_scene.backend.check(fn(oracle_scene._h, ids.ctypes.data if len(ids) else None, len(ids)))


def tile_mask(width, height, tile_ids):
    """Boolean [H, W] mask of the pixels of the listed tiles."""
    tiles_x = (width + 15) // 16
    m = np.zeros((height, width), bool)
    for t in tile_ids:
        ty, tx = divmod(int(t), tiles_x)
        m[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = True
    return m
