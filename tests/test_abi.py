"""The drop-in boundary: both libraries export every symbol include/jade_rt.h declares,
the ctypes mirrors match the C layout, and the product fails loudly without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import B, J, ROOT, config_scene
from jaderaytracerendering_amd import _abi


def _declared_symbols(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(jadeh?_[a-z0-9_]+)\s*\(", text))


def test_python_tables_cover_the_headers():
    assert _declared_symbols("jade_rt.h") == set(_abi.RT_SYMBOLS)
    assert _declared_symbols("jade_host_c.h") == set(_abi.HOST_SYMBOLS)
    assert _declared_symbols("jade_bvh.h") == set(_abi.BVH_SYMBOLS)


@pytest.mark.parametrize("which", ["hip", "oracle"])
def test_library_loads_and_exports_every_symbol(which):
    path = B.HIP_LIB if which == "hip" else os.path.join(ROOT, "oracle", "libjade_oracle.so")
    assert os.path.exists(path), f"{path} not built (run __graft_entry__.build())"
    lib = ctypes.CDLL(path)
    for name in _abi.RT_SYMBOLS:
        assert hasattr(lib, name), f"{path} lacks {name}"
    _abi.bind(lib, _abi.RT_SYMBOLS)
    if which == "hip":
        _abi.bind(lib, _abi.BVH_SYMBOLS)   # include/jade_bvh.h lives in the HIP library only
    assert lib.jade_abi_version() == _abi.JADE_ABI_VERSION
    assert lib.jade_backend_name().decode() == ("hip-gfx950" if which == "hip" else "oracle-cpu")
    assert lib.jade_owned_tile_count(1920, 1080, 0, 8) == 1020  # 120 x 68 tiles, diagonal deal
    assert [lib.jade_owned_tile_count(33, 17, r, 4) for r in range(4)] == [1, 2, 2, 1]  # 3x2 tiles, (tx+ty) % 4
    assert lib.jade_owned_tile_count(0, 17, 0, 1) == -1


def test_struct_layout_matches_the_c_compiler(fpm):
    out = (ctypes.c_int * 14)()
    fpm.t_layout(out)
    # 112 / 40 / 8: sizeof Triangle_cu / BVHNode_cu / Obj_seg measured in SURVEY.md section 2 row 8
    assert list(out[:3]) == [112, 40, 8]
    assert list(out[:6]) == [ctypes.sizeof(t) for t in (_abi.Triangle, _abi.BvhNode, _abi.ObjSeg, _abi.SceneDesc,
                                                          _abi.RenderParams, _abi.Stats)]
    # offsets SURVEY.md lists: norm@40 emissive@52 brdf@64 reflex@76 rate@84 albedo@96 index@108
    assert list(out[6:13]) == [40, 52, 64, 76, 84, 96, 108]
    assert [getattr(_abi.Triangle, f).offset for f in ("norm", "emissive", "brdf", "reflex_mode", "refract_rate",
                                                       "refract_albedo", "refract_index")] == list(out[6:13])
    assert out[13] == _abi.BvhNode.aa.offset == 16


def test_product_fails_loudly_without_a_gpu():
    """No CPU fallback: on a box without a HIP device the product returns an error status."""
    hip = J.hip()
    try:
        n = hip.device_count()
    except B.JadeError as e:
        assert e.code == _abi.JADE_ERR_DEVICE
        n = 0
    if n > 0:
        pytest.skip("a GPU is present; the no-device path is not reachable")
    hs, _ = config_scene("tiny")
    with pytest.raises(B.JadeError) as ei:
        hip.scene(hs)
    assert ei.value.code == _abi.JADE_ERR_DEVICE


@pytest.mark.parametrize("which", ["oracle", "hip"])
def test_scene_validation_errors(which, oracle):
    """Bad arrays are rejected with a status + message, never a crash (no exit() across the ABI)."""
    be = oracle if which == "oracle" else J.hip()
    hs, _ = config_scene("tiny")

    def expect(mutator, codes):
        bad = J.HostScene({k: v.copy() for k, v in hs.a.items()})
        mutator(bad)
        with pytest.raises(B.JadeError) as ei:
            be.scene(bad)
        assert ei.value.code in codes and str(ei.value)

    inv = (_abi.JADE_ERR_INVALID, _abi.JADE_ERR_UNSUPPORTED)
    expect(lambda s: s.a["emit"].__setitem__(0, 10 ** 6), inv)
    expect(lambda s: s.a["mapping"].__setitem__(0, -5), inv)
    expect(lambda s: s.a["segs"].__setitem__((0, 1), 10 ** 6), inv)
    expect(lambda s: s.node_i32().__setitem__((1, 0), 10 ** 6), inv)     # child out of range
    expect(lambda s: s.node_i32().__setitem__((1, 0), 1), inv)            # cycle: root is its own child

    def deep_chain(s):  # a BVH deeper than the 128-entry traversal stack
        n = 200
        nodes = np.zeros((2 * n + 2, 10), np.uint32)
        ni = nodes.view(np.int32)
        for i in range(1, n + 1):
            ni[i, 0] = i + 1 if i < n else n + 1
            ni[i, 1] = n + 1 + i if i < n else 2 * n + 1
        for j in list(range(n + 1, 2 * n + 2)):
            ni[j, 2], ni[j, 3] = 1, 0
        ni[n, 0], ni[n, 1] = n + 1, 2 * n + 1
        s.a["nodes"] = nodes
    expect(deep_chain, (_abi.JADE_ERR_UNSUPPORTED,))


def test_abi_version_is_single_sourced():
    """include/jade_rt.h, the Python mirror and both libraries agree (a stale binary must not pass)."""
    text = open(os.path.join(ROOT, "include", "jade_rt.h")).read()
    ver = int(re.search(r"#define JADE_ABI_VERSION (\d+)", text).group(1))
    assert ver == _abi.JADE_ABI_VERSION
    assert int(re.search(r"#define JADE_SAMPLE_LANES (\d+)", text).group(1)) == _abi.JADE_SAMPLE_LANES
    for path in (B.HIP_LIB, os.path.join(ROOT, "oracle", "libjade_oracle.so")):
        assert ctypes.CDLL(path).jade_abi_version() == ver, path


def test_product_library_exports_only_the_boundary():
    """libjade_hip.so's dynamic symbols named jade_* are exactly what include/jade_rt.h and include/jade_bvh.h declare: the
    jade_debug_* entry points the piece-by-piece GPU tests use live in libjade_hip_debug.so."""
    import subprocess
    path = os.path.join(ROOT, "jaderaytracerendering_amd", "lib", "libjade_hip.so")
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.split()[-1].startswith("jade_")}
    declared = _declared_symbols("jade_rt.h") | _declared_symbols("jade_bvh.h")
    assert exported - declared <= {"jade_fail"}, sorted(exported - declared)  # (jade_fail: shared by the module's two translation units)
    assert declared <= exported
    dbg = os.path.join(ROOT, "jaderaytracerendering_amd", "lib", "libjade_hip_debug.so")
    out = subprocess.run(["nm", "-D", "--defined-only", dbg], capture_output=True, text=True, check=True).stdout
    assert "jade_debug_trace_rays_limit" in out and "jade_debug_trace_rays_cached" in out
