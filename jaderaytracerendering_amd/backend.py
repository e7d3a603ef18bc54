"""Binding of the jade_rt.h C ABI.

`hip()` returns the product backend (libjade_hip.so, hand-written HIP for
gfx950) and raises if it is missing — there is no CPU fallback in the product
path.  `Backend(path)` binds any library implementing jade_rt.h; the test
suite uses it to load the CPU oracle as the checker.
"""
import ctypes as C
import os

import numpy as np

from . import _abi

_LIBDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
HIP_LIB = os.path.join(_LIBDIR, "libjade_hip.so")


class JadeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"jade_rt status {code}: {msg}")
        self.code = code


def make_params(width, height, spp, eye, camera, frame=0, tile_rank=0, tile_nranks=1, device_id=0, threads=0, walk=_abi.WALK_REFERENCE, env_sampling=_abi.ENV_REFERENCE):
    p = _abi.RenderParams()
    p.width, p.height, p.spp, p.frame = int(width), int(height), int(spp), int(frame)
    p.eye[:] = [float(v) for v in eye]
    p.camera[:] = [float(v) for v in camera]
    p.tile_rank, p.tile_nranks = int(tile_rank), int(tile_nranks)
    p.device_id, p.threads = int(device_id), int(threads)
    p.walk = int(walk)  # jade_rt.h, JADE_WALK_*: 0 = the reference's walk (V / T equal the oracle's), 1 = early exits, 2 = + occluder cache
    p.env_sampling = int(env_sampling)  # JADE_ENV_*: 0 = the reference's estimator; 1 = environment rays by importance (non-parity)
    return p


def params_from_config(cfg, **kw):
    kw.setdefault("spp", cfg.spp)
    return make_params(cfg.width, cfg.height, kw.pop("spp"), list(cfg.eye), list(cfg.camera), **kw)


class Backend:
    def __init__(self, path):
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it first (make / __graft_entry__.build())")
        self.path = path
        self.lib = _abi.bind(C.CDLL(path), _abi.RT_SYMBOLS)
        if self.lib.jade_abi_version() != _abi.JADE_ABI_VERSION:
            raise RuntimeError(f"{path}: ABI version mismatch")

    @property
    def name(self):
        return self.lib.jade_backend_name().decode()

    def check(self, rc):
        if rc != 0:
            raise JadeError(rc, self.lib.jade_last_error().decode())

    def device_count(self):
        n = C.c_int(0)
        self.check(self.lib.jade_device_count(C.byref(n)))
        return n.value

    def scene(self, host_scene, device_id=0):
        return Scene(self, host_scene, device_id)

    def owned_tile_count(self, width, height, rank, nranks):
        return self.lib.jade_owned_tile_count(width, height, rank, nranks)


class Scene:
    """A scene resident on the backend (PathTrace.cu:1618-1698 on the reference side)."""

    def __init__(self, backend, host_scene, device_id=0):
        self.backend = backend
        self.host_scene = host_scene
        self._h = C.c_void_p()
        desc = host_scene.desc()
        backend.check(backend.lib.jade_scene_create(C.byref(desc), device_id, C.byref(self._h)))
        self._params = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.backend.lib.jade_scene_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def render(self, params, want_rgb=True, want_bgr8=True):
        """(rgb float32 [H,W,3] | None, bgr8 uint8 [H,W,3] | None, Stats); row 0 = bottom row."""
        h, w = params.height, params.width
        rgb = np.zeros((h, w, 3), np.float32) if want_rgb else None
        bgr = np.zeros((h, w, 3), np.uint8) if want_bgr8 else None
        st = _abi.Stats()
        self.backend.check(self.backend.lib.jade_render(self._h, C.byref(params), rgb.ctypes.data if want_rgb else None,
                                                        bgr.ctypes.data if want_bgr8 else None, C.byref(st)))
        return rgb, bgr, st

    # progressive form
    def begin(self, params):
        self._params = params
        self.backend.check(self.backend.lib.jade_render_begin(self._h, C.byref(params)))

    def step(self, spp, stats=None):
        st = stats if stats is not None else _abi.Stats()
        self.backend.check(self.backend.lib.jade_render_step(self._h, int(spp), C.byref(st)))
        return st

    def flush(self, stats=None):
        """Finish the paths a step may have carried over (jade_rt.h); resolve() does it implicitly."""
        st = stats if stats is not None else _abi.Stats()
        self.backend.check(self.backend.lib.jade_render_flush(self._h, C.byref(st)))
        return st

    def resolve(self, want_rgb=True, want_bgr8=True, tonemap=None, limit=1.5):
        """tonemap None/ACES: PathTrace.cu:680-682; _abi.TONEMAP_REINHARD: the preview's pass3.fsh operator."""
        h, w = self._params.height, self._params.width
        rgb = np.zeros((h, w, 3), np.float32) if want_rgb else None
        bgr = np.zeros((h, w, 3), np.uint8) if want_bgr8 else None
        pr, pb = (rgb.ctypes.data if want_rgb else None), (bgr.ctypes.data if want_bgr8 else None)
        if tonemap is None:
            self.backend.check(self.backend.lib.jade_render_resolve(self._h, pr, pb))
        else:
            self.backend.check(self.backend.lib.jade_render_resolve_ex(self._h, int(tonemap), float(limit), pr, pb))
        return rgb, bgr

    def query(self, what):
        """jade_render_query: what the backend holds for the current render (_abi.Q_*)."""
        v = C.c_int64(0)
        self.backend.check(self.backend.lib.jade_render_query(self._h, int(what), C.byref(v)))
        return v.value

    def resolve_tiles_device(self, dev_ptr, stream=0):
        self.backend.check(self.backend.lib.jade_render_resolve_tiles_device(self._h, C.c_void_p(dev_ptr), C.c_void_p(stream)))

    def trace_rays(self, origins, dirs, skip):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        s = np.ascontiguousarray(skip, np.int32).reshape(-1)
        n = len(o)
        assert len(d) == n and len(s) == n
        idx = np.zeros(n, np.int32)
        dist = np.zeros(n, np.float32)
        pt = np.zeros((n, 3), np.float32)
        st = _abi.Stats()
        self.backend.check(self.backend.lib.jade_trace_rays(self._h, n, o.ctypes.data, d.ctypes.data, s.ctypes.data,
                                                            idx.ctypes.data, dist.ctypes.data, pt.ctypes.data, C.byref(st)))
        return idx, dist, pt, st


def render_multi(backend, scenes, params, want_rgb=True, want_bgr8=True):
    """jade_render_multi: one frame over several scenes (one per device) from this process."""
    h, w = params.height, params.width
    rgb = np.zeros((h, w, 3), np.float32) if want_rgb else None
    bgr = np.zeros((h, w, 3), np.uint8) if want_bgr8 else None
    st = _abi.Stats()
    arr = (C.c_void_p * len(scenes))(*[s._h for s in scenes])
    backend.check(backend.lib.jade_render_multi(arr, len(scenes), C.byref(params), rgb.ctypes.data if want_rgb else None,
                                                bgr.ctypes.data if want_bgr8 else None, C.byref(st)))
    return rgb, bgr, st


_hip = None


def hip():
    """The product backend.  Fails loudly if the HIP extension was not built."""
    global _hip
    if _hip is None:
        # JADE_HIP_LIB points development tools at another BUILD of the same HIP library (A/B runs)
        _hip = Backend(os.environ.get("JADE_HIP_LIB", HIP_LIB))
    return _hip


def assemble_tiles(tiles, width, height, rank, nranks, out=None):
    """Scatter one rank's compact tile buffer ([n_owned, 16, 16, 3]) into a full image."""
    ts = _abi.TILE_SIZE
    tx = (width + ts - 1) // ts
    ty = (height + ts - 1) // ts
    if out is None:
        out = np.zeros((height, width, 3), tiles.dtype)
    from .distributed import owned_tile_ids
    ids = owned_tile_ids(width, height, rank, nranks)
    for k, tid in enumerate(ids):
        y0, x0 = (tid // tx) * ts, (tid % tx) * ts
        hh, ww = min(ts, height - y0), min(ts, width - x0)
        out[y0:y0 + hh, x0:x0 + ww] = tiles[k, :hh, :ww]
    return out
