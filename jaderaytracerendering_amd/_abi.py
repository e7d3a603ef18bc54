"""ctypes mirrors of include/jade_rt.h and include/jade_host_c.h.

Field order and types must match the headers exactly; tests/test_abi.py checks
the struct sizes against the C compiler's (112 / 40 / 8 bytes for the three
reference device structs, PathTrace.cu:327-351).
"""
import ctypes as C

JADE_ABI_VERSION = 7
JADE_SAMPLE_LANES = 1024
JADE_OK, JADE_ERR_INVALID, JADE_ERR_DEVICE, JADE_ERR_NOMEM, JADE_ERR_UNSUPPORTED = range(5)
DIFFUSE, MIRROR = 0, 1
NO_REFRACT, SUB_SURFACE, DIR_REFRACT = 0, 1, 2
TILE_SIZE = 16
TONEMAP_ACES, TONEMAP_REINHARD = 0, 1
Q_RECORDS_PER_PIXEL, Q_STATE_BYTES, Q_SUM_LANES = 0, 1, 2
WALK_REFERENCE, WALK_EARLY_EXIT, WALK_EARLY_EXIT_CACHED = 0, 1, 2
ENV_REFERENCE, ENV_IMPORTANCE = 0, 1

f3 = C.c_float * 3
f16 = C.c_float * 16


class Triangle(C.Structure):  # == Triangle_cu, PathTrace.cu:327-338
    _fields_ = [
        ("obj_idx", C.c_int32),
        ("p1", f3), ("p2", f3), ("p3", f3),
        ("norm", f3), ("emissive", f3), ("brdf", f3),
        ("reflex_mode", C.c_int32), ("refract_mode", C.c_int32),
        ("refract_rate", f3), ("refract_albedo", f3),
        ("refract_index", C.c_float),
    ]


class BvhNode(C.Structure):  # == BVHNode_cu, PathTrace.cu:341-345
    _fields_ = [("left", C.c_int32), ("right", C.c_int32), ("n", C.c_int32), ("index", C.c_int32),
                ("aa", f3), ("bb", f3)]


class ObjSeg(C.Structure):  # == Obj_seg, PathTrace.cu:348-351
    _fields_ = [("begin_idx", C.c_int32), ("end_idx", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32),
        ("n_triangles", C.c_int32), ("triangles", C.POINTER(Triangle)),
        ("n_nodes", C.c_int32), ("nodes", C.POINTER(BvhNode)),
        ("n_emit", C.c_int32), ("emit_indices", C.POINTER(C.c_int32)),
        ("index_mapping", C.POINTER(C.c_int32)),
        ("prefix_area", C.POINTER(C.c_float)),
        ("n_objects", C.c_int32), ("obj_segs", C.POINTER(ObjSeg)),
        ("env_width", C.c_int32), ("env_height", C.c_int32),
        ("env_rgb", C.POINTER(C.c_float)),
    ]


class RenderParams(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("frame", C.c_uint32),
        ("eye", f3), ("camera", f16),
        ("tile_rank", C.c_int32), ("tile_nranks", C.c_int32),
        ("device_id", C.c_int32), ("threads", C.c_int32),
        ("max_state_bytes", C.c_uint64),
        ("walk", C.c_int32),
        ("env_sampling", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("rays_primary", C.c_uint64), ("rays_secondary", C.c_uint64),
        ("nodes_visited", C.c_uint64), ("tris_tested", C.c_uint64),
        ("shaded_hits", C.c_uint64), ("samples", C.c_uint64),
        ("kernel_ms", C.c_double),
        ("trace_ms", C.c_double), ("trace_launches", C.c_uint64),
        ("rays_shadow", C.c_uint64), ("rays_env", C.c_uint64), ("rays_indirect", C.c_uint64),
        ("rays_mirror", C.c_uint64), ("rays_refract", C.c_uint64), ("host_syncs", C.c_uint64),
        ("rays_inline", C.c_uint64), ("light_ms", C.c_double),
        ("nodes_inline", C.c_uint64), ("tris_inline", C.c_uint64),
        ("rays_cached", C.c_uint64),
        ("rays_tail", C.c_uint64), ("nodes_tail", C.c_uint64), ("tris_tail", C.c_uint64),
        ("tail_ms", C.c_double), ("tail_launches", C.c_uint64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}

    @property
    def rays(self):
        return self.rays_primary + self.rays_secondary


class Material(C.Structure):  # == Material, PathTrace.cu:293-301
    _fields_ = [
        ("emissive", f3), ("brdf", f3),
        ("reflex_mode", C.c_int32), ("refract_mode", C.c_int32),
        ("refract_rate", f3), ("refract_albedo", f3),
        ("refract_index", C.c_float),
    ]


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp", C.c_int32), ("eye", f3), ("camera", f16)]


# every symbol jade_rt.h declares: name -> (restype, argtypes)
RT_SYMBOLS = {
    "jade_abi_version": (C.c_int, []),
    "jade_backend_name": (C.c_char_p, []),
    "jade_last_error": (C.c_char_p, []),
    "jade_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "jade_scene_create": (C.c_int, [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "jade_scene_destroy": (None, [C.c_void_p]),
    "jade_render": (C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.POINTER(Stats)]),
    "jade_render_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.POINTER(RenderParams), C.c_void_p, C.c_void_p,
                                    C.POINTER(Stats)]),
    "jade_render_begin": (C.c_int, [C.c_void_p, C.POINTER(RenderParams)]),
    "jade_render_step": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(Stats)]),
    "jade_render_flush": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "jade_render_resolve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "jade_render_resolve_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "jade_render_resolve_tiles_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "jade_render_query": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]),
    "jade_owned_tile_count": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "jade_trace_rays": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.POINTER(Stats)]),
}

# include/jade_bvh.h (exported by libjade_hip.so)
_BVH_SIG = (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32),
                      C.POINTER(C.c_double)])
BVH_SYMBOLS = {"jade_bvh_build_lbvh": _BVH_SIG, "jade_bvh_build_ploc": _BVH_SIG}

HOST_SYMBOLS = {
    "jadeh_last_error": (C.c_char_p, []),
    "jadeh_builder_new": (C.c_void_p, []),
    "jadeh_builder_free": (None, [C.c_void_p]),
    "jadeh_builder_triangle_count": (C.c_int, [C.c_void_p]),
    "jadeh_builder_add_mesh": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(Material),
                                         C.c_void_p, C.c_int]),
    "jadeh_builder_add_obj": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(Material), C.c_void_p, C.c_int]),
    "jadeh_builder_add_proc": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_uint, C.POINTER(Material), C.c_void_p,
                                         C.c_int]),
    "jadeh_write_proc_obj": (C.c_int, [C.c_char_p, C.c_int, C.c_uint, C.c_char_p]),
    "jadeh_builder_set_env_constant": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float]),
    "jadeh_builder_set_env_sky": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "jadeh_builder_set_env_data": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "jadeh_builder_set_env_hdr": (C.c_int, [C.c_void_p, C.c_char_p]),
    "jadeh_builder_config": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(Config)]),
    "jadeh_builder_load_render_args": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(Config)]),
    "jadeh_builder_build": (C.c_void_p, [C.c_void_p, C.c_int]),
    "jadeh_builder_triangles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "jadeh_builder_build_with_bvh": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "jadeh_scene_free": (None, [C.c_void_p]),
    "jadeh_scene_desc": (None, [C.c_void_p, C.POINTER(SceneDesc)]),
    "jadeh_scene_bvh_depth": (C.c_int, [C.c_void_p]),
    "jadeh_scene_build_seconds": (C.c_double, [C.c_void_p]),
    "jadeh_transform_matrix": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jadeh_camera_orbit": (None, [C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "jadeh_write_bmp": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int]),
    "jadeh_write_ppm": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int]),
    "jadeh_write_pfm": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int]),
}


def bind(lib, table):
    """Attach restype/argtypes; raises AttributeError naming a missing symbol."""
    for name, (res, args) in table.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib
