"""Python face of the host scene pipeline (libjade_host.so, C++).

Roles of PathTrace.cu:355-628, 1487-1612 and PathTrace.cpp:343-359, 684-687:
mesh loading / procedural stand-ins, SAH BVH, flattening, camera.  The heavy
lifting is native; this module only marshals.
"""
import ctypes as C
import os

import numpy as np

from . import _abi

_LIBDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
_host = None


def host_lib():
    global _host
    if _host is None:
        path = os.path.join(_LIBDIR, "libjade_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make host` (or __graft_entry__.build())")
        _host = _abi.bind(C.CDLL(path), _abi.HOST_SYMBOLS)
    return _host


def _check(rc):
    if rc != 0:
        raise RuntimeError(host_lib().jadeh_last_error().decode())


def material(emissive=(0, 0, 0), brdf=(0.8, 0.8, 0.8), reflex_mode=_abi.DIFFUSE, refract_mode=_abi.NO_REFRACT,
             refract_rate=(0.8, 0.8, 0.8), refract_albedo=(0.8, 0.8, 0.8), refract_index=1.0):
    m = _abi.Material()
    m.emissive[:] = emissive
    m.brdf[:] = brdf
    m.reflex_mode = reflex_mode
    m.refract_mode = refract_mode
    m.refract_rate[:] = refract_rate
    m.refract_albedo[:] = refract_albedo
    m.refract_index = refract_index
    return m


def jade_material():
    """The reference's jade: PathTrace.cpp:981-989."""
    return material(brdf=(0.02,) * 3, reflex_mode=_abi.MIRROR, refract_mode=_abi.SUB_SURFACE, refract_rate=(0.1,) * 3,
                    refract_albedo=(0.3,) * 3, refract_index=2.66)


def transform_matrix(rot_deg=(0, 0, 0), trans=(0, 0, 0), scale=(1, 1, 1)):
    """getTransformMatrix (PathTrace.cpp:343-359): T * Rx * Ry * Rz * S, [col][row] order."""
    out = np.zeros(16, np.float32)
    r, t, s = (np.asarray(v, np.float32) for v in (rot_deg, trans, scale))
    host_lib().jadeh_transform_matrix(r.ctypes.data, t.ctypes.data, s.ctypes.data, out.ctypes.data)
    return out


def camera_orbit(r=4.0, up_deg=0.0, rot_deg=0.0, center=(0, 0, 0)):
    """(eye[3], camera[16]) as the GL host computes them (PathTrace.cpp:684-687)."""
    eye = np.zeros(3, np.float32)
    cam = np.zeros(16, np.float32)
    c = np.asarray(center, np.float32)
    host_lib().jadeh_camera_orbit(r, up_deg, rot_deg, c.ctypes.data, eye.ctypes.data, cam.ctypes.data)
    return eye, cam


class HostScene:
    """Flattened scene: the arrays that cross the jade_rt.h boundary."""

    def __init__(self, arrays, bvh_depth=None, build_seconds=0.0):
        # arrays: dict of numpy arrays (owned copies)
        self.a = arrays
        self.bvh_depth = bvh_depth
        self.build_seconds = build_seconds

    ARRAY_KEYS = ("triangles", "nodes", "emit", "mapping", "prefix", "segs", "env")

    @classmethod
    def from_handle(cls, handle):
        lib = host_lib()
        d = _abi.SceneDesc()
        lib.jadeh_scene_desc(handle, C.byref(d))

        def grab(ptr, n, dtype):
            if n == 0:
                return np.zeros(0, dtype)
            buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(C.addressof(ptr.contents))
            return np.frombuffer(buf, dtype=dtype).copy()

        arrays = {
            "triangles": grab(d.triangles, d.n_triangles * 28, np.uint32).reshape(-1, 28),
            "nodes": grab(d.nodes, d.n_nodes * 10, np.uint32).reshape(-1, 10),
            "emit": grab(d.emit_indices, d.n_emit, np.int32),
            "mapping": grab(d.index_mapping, d.n_triangles, np.int32),
            "prefix": grab(d.prefix_area, d.n_triangles, np.float32),
            "segs": grab(d.obj_segs, d.n_objects * 2, np.int32).reshape(-1, 2),
            "env": grab(d.env_rgb, d.env_width * d.env_height * 3, np.float32).reshape(d.env_height, d.env_width, 3),
        }
        return cls(arrays, lib.jadeh_scene_bvh_depth(handle), lib.jadeh_scene_build_seconds(handle))

    @classmethod
    def from_npz(cls, path):
        with np.load(path, allow_pickle=False) as z:
            return cls({k: z[k].copy() for k in cls.ARRAY_KEYS})

    def save_npz(self, path, **extra):
        np.savez_compressed(path, **self.a, **extra)

    @property
    def n_triangles(self):
        return self.a["triangles"].shape[0]

    @property
    def n_nodes(self):
        return self.a["nodes"].shape[0]

    # structured views (fields of Triangle_cu / BVHNode_cu)
    def tri_f32(self):
        return self.a["triangles"].view(np.float32)

    def tri_i32(self):
        return self.a["triangles"].view(np.int32)

    def vertices(self):
        """(nT, 3, 3) float32: p1, p2, p3."""
        return self.tri_f32()[:, 1:10].reshape(-1, 3, 3)

    def node_i32(self):
        return self.a["nodes"].view(np.int32)

    def node_f32(self):
        return self.a["nodes"].view(np.float32)

    def desc(self):
        """A SceneDesc pointing into this object's arrays (keep `self` alive)."""
        a = self.a
        for k in self.ARRAY_KEYS:
            a[k] = np.ascontiguousarray(a[k])
        d = _abi.SceneDesc()
        d.abi_version = _abi.JADE_ABI_VERSION
        d.n_triangles = a["triangles"].shape[0]
        d.triangles = C.cast(a["triangles"].ctypes.data, C.POINTER(_abi.Triangle))
        d.n_nodes = a["nodes"].shape[0]
        d.nodes = C.cast(a["nodes"].ctypes.data, C.POINTER(_abi.BvhNode))
        d.n_emit = a["emit"].shape[0]
        d.emit_indices = C.cast(a["emit"].ctypes.data, C.POINTER(C.c_int32))
        d.index_mapping = C.cast(a["mapping"].ctypes.data, C.POINTER(C.c_int32))
        d.prefix_area = C.cast(a["prefix"].ctypes.data, C.POINTER(C.c_float))
        d.n_objects = a["segs"].shape[0]
        d.obj_segs = C.cast(a["segs"].ctypes.data, C.POINTER(_abi.ObjSeg))
        d.env_height, d.env_width = a["env"].shape[:2]
        d.env_rgb = C.cast(a["env"].ctypes.data, C.POINTER(C.c_float))
        return d


class SceneBuilder:
    """One add_*() call == one readObj() call of the reference (one object)."""

    def __init__(self):
        self._lib = host_lib()
        self._h = self._lib.jadeh_builder_new()

    def close(self):
        if self._h:
            self._lib.jadeh_builder_free(self._h)
            self._h = None

    __del__ = close

    @staticmethod
    def _t(trans):
        if trans is None:
            trans = transform_matrix()
        return np.ascontiguousarray(trans, np.float32)

    def add_mesh(self, vertices, indices, mat, trans=None, normalize=False):
        v = np.ascontiguousarray(vertices, np.float32).reshape(-1, 3)
        i = np.ascontiguousarray(indices, np.int32).reshape(-1, 3)
        t = self._t(trans)
        _check(self._lib.jadeh_builder_add_mesh(self._h, v.ctypes.data, len(v), i.ctypes.data, len(i), C.byref(mat),
                                                t.ctypes.data, int(normalize)))

    def add_obj(self, path, mat, trans=None, normalize=False):
        t = self._t(trans)
        _check(self._lib.jadeh_builder_add_obj(self._h, os.fsencode(path), C.byref(mat), t.ctypes.data, int(normalize)))

    def add_proc(self, kind, param, mat, trans=None, normalize=False, seed=0):
        t = self._t(trans)
        _check(self._lib.jadeh_builder_add_proc(self._h, kind.encode(), int(param), int(seed), C.byref(mat),
                                                t.ctypes.data, int(normalize)))

    def set_env_constant(self, r, g, b):
        _check(self._lib.jadeh_builder_set_env_constant(self._h, r, g, b))

    def set_env_sky(self, w=1024, h=512):
        _check(self._lib.jadeh_builder_set_env_sky(self._h, w, h))

    def set_env_data(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.float32)
        h, w = rgb.shape[:2]
        _check(self._lib.jadeh_builder_set_env_data(self._h, w, h, rgb.ctypes.data))

    def set_env_hdr(self, path):
        _check(self._lib.jadeh_builder_set_env_hdr(self._h, os.fsencode(path)))

    def config(self, name):
        """Fill the builder with a built-in configuration; returns its Config."""
        cfg = _abi.Config()
        _check(self._lib.jadeh_builder_config(self._h, name.encode(), C.byref(cfg)))
        return cfg

    def load_render_args(self, path):
        cfg = _abi.Config()
        _check(self._lib.jadeh_builder_load_render_args(self._h, os.fsencode(path), C.byref(cfg)))
        return cfg

    @property
    def triangle_count(self):
        return self._lib.jadeh_builder_triangle_count(self._h)

    def triangles_original(self):
        """(n, 28) uint32 view of the Triangle_cu records in ORIGINAL order (input of an external BVH builder)."""
        n = self.triangle_count
        out = np.zeros((n, 28), np.uint32)
        _check(self._lib.jadeh_builder_triangles(self._h, out.ctypes.data, n))
        return out

    def build_with_bvh(self, order, nodes):
        """Flatten around a BVH built elsewhere: order[i] = original index of sorted triangle i."""
        order = np.ascontiguousarray(order, np.int32)
        nodes = np.ascontiguousarray(nodes, np.uint32).reshape(-1, 10)
        h = self._lib.jadeh_builder_build_with_bvh(self._h, order.ctypes.data, nodes.ctypes.data, len(nodes))
        if not h:
            raise RuntimeError(self._lib.jadeh_last_error().decode())
        try:
            return HostScene.from_handle(h)
        finally:
            self._lib.jadeh_scene_free(h)

    def build_device_bvh(self, backend, kind="ploc", leaf_size=8, device_id=0):
        """(HostScene, device build milliseconds) with the BVH built on the GPU (include/jade_bvh.h): "lbvh" (Morton order +
        Karras' radix tree) or "ploc" (locally-ordered clustering by surface area)."""
        if kind not in ("lbvh", "ploc"):
            raise ValueError(f"unknown device BVH builder {kind!r}")
        tris = self.triangles_original()
        n = len(tris)
        lib = _abi.bind(backend.lib, _abi.BVH_SYMBOLS)
        order = np.zeros(n, np.int32)
        nodes = np.zeros((2 * n + 1, 10), np.uint32)
        n_nodes = C.c_int32(0)
        ms = C.c_double(0)
        fn = lib.jade_bvh_build_lbvh if kind == "lbvh" else lib.jade_bvh_build_ploc
        backend.check(fn(tris.ctypes.data, n, leaf_size, device_id, order.ctypes.data, nodes.ctypes.data, len(nodes), C.byref(n_nodes),
                         C.byref(ms)))
        return self.build_with_bvh(order, nodes[: n_nodes.value]), ms.value

    def build_lbvh(self, backend, leaf_size=8, device_id=0):
        return self.build_device_bvh(backend, "lbvh", leaf_size, device_id)

    def build_ploc(self, backend, leaf_size=3, device_id=0):
        # 3 triangles per leaf: measured on C3 / C5 the best trade of node records against triangle tests for this builder
        return self.build_device_bvh(backend, "ploc", leaf_size, device_id)

    def build(self, leaf_size=8):
        h = self._lib.jadeh_builder_build(self._h, leaf_size)
        if not h:
            raise RuntimeError(self._lib.jadeh_last_error().decode())
        try:
            return HostScene.from_handle(h)
        finally:
            self._lib.jadeh_scene_free(h)


def build_config(name):
    """(HostScene, Config) for a built-in configuration: tiny, tinyjade, C1..C5."""
    b = SceneBuilder()
    try:
        cfg = b.config(name)
        return b.build(), cfg
    finally:
        b.close()


def object_tiles(hs, eye, camera, width, height, obj=0):
    """{tile id (ty * tiles_x + tx): vertices of object `obj` that project into it}: the camera mapping of
    PathTrace.cu:1430-1437 inverted (dir ~ M . (lx * W/H, ly, -1.5, 0), lx = -1 + 2/W (x + u - 0.5), M[col][row])."""
    v = hs.vertices()[hs.tri_i32()[:, 0] == obj].reshape(-1, 3).astype(np.float64)
    m = np.asarray(list(camera), np.float64).reshape(4, 4)
    rel = v - np.asarray(list(eye), np.float64)
    a, b, c = rel @ m[0, :3], rel @ m[1, :3], rel @ m[2, :3]
    front = c < 0
    s = -1.5 / c[front]
    lx, ly = a[front] * s / (width / height), b[front] * s
    x, y = np.floor((lx + 1) * width / 2), np.floor((ly + 1) * height / 2)
    ok = (x >= 0) & (x < width) & (y >= 0) & (y < height)
    tiles_x = (width + 15) // 16
    ids, cnt = np.unique((y[ok] // 16).astype(np.int64) * tiles_x + (x[ok] // 16).astype(np.int64), return_counts=True)
    return dict(zip(ids.tolist(), cnt.tolist()))


def write_bmp(path, bgr8):
    h, w = bgr8.shape[:2]
    a = np.ascontiguousarray(bgr8, np.uint8)
    _check(host_lib().jadeh_write_bmp(os.fsencode(path), a.ctypes.data, w, h))


def write_ppm(path, bgr8):
    h, w = bgr8.shape[:2]
    a = np.ascontiguousarray(bgr8, np.uint8)
    _check(host_lib().jadeh_write_ppm(os.fsencode(path), a.ctypes.data, w, h))


def write_pfm(path, rgb):
    h, w = rgb.shape[:2]
    a = np.ascontiguousarray(rgb, np.float32)
    _check(host_lib().jadeh_write_pfm(os.fsencode(path), a.ctypes.data, w, h))
