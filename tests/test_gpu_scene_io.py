"""Scene I/O through the GPU: a render_args.txt scene made of .obj files and an RGBE .hdr
environment, rendered by the repo's own C++ front end (jade_render) with libjade_hip.so and checked against the same
command with the oracle as the backend.  Reference side: the scene file reader PathTrace.cu:1487-1525, readObj :355-457
(slashes, normalisation quirk), the HDR upload :1648-1689 (hdrloader is un-vendored: own RGBE reader), save_image :74-106."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ORACLE_LIB, ROOT, rel_l2
from jaderaytracerendering_amd import host as H
from jaderaytracerendering_amd.backend import HIP_LIB
from test_host_pipeline import _write_hdr

pytestmark = pytest.mark.gpu
EXE = os.path.join(ROOT, "jaderaytracerendering_amd", "lib", "jade_render")


def _mat4(t):
    return [" ".join(f"{x:.9g}" for x in t[4 * c:4 * c + 4]) for c in range(4)]


def _obj(path, verts, faces, slashes=False):
    with open(path, "w") as f:
        f.write("# test mesh\n")
        for v in verts:
            f.write("v %.9g %.9g %.9g\n" % tuple(v))
        for a, b, c in faces:
            f.write(("f %d/1/1 %d/1/1 %d/1/1\n" if slashes else "f %d %d %d\n") % (a + 1, b + 1, c + 1))


def _read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = (int(x) for x in f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(), "<f4" if scale < 0 else ">f4").reshape(h, w, 3)
    return data


def _run(backend, args, env, out, size, spp, extra=()):
    r = subprocess.run([EXE, "--args", args, "--env", env, "--width", str(size[0]), "--height", str(size[1]), "--spp", str(spp),
                        "--backend", backend, "--out", out] + list(extra), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:] + r.stdout[-500:]
    stats = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    return stats, r.stdout


def test_render_args_obj_hdr_scene_hip_vs_oracle(tmp_path):
    subprocess.check_call(["make", "-s", "-C", ROOT, "cli"])
    d = str(tmp_path)
    lib = H.host_lib()
    assert lib.jadeh_write_proc_obj(b"statue", 9, 5, os.path.join(d, "statue.obj").encode()) == 0      # 1620 triangles
    assert lib.jadeh_write_proc_obj(b"geodesic", 3, 0, os.path.join(d, "ball.obj").encode()) == 0       # 180 triangles
    _obj(os.path.join(d, "light.obj"), [[-0.5, -0.5, 0], [0.5, -0.5, 0], [0.5, 0.5, 0], [-0.5, 0.5, 0]], [[0, 1, 2], [0, 2, 3]], slashes=True)
    box_v = [[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)]
    box_f = [[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]]
    _obj(os.path.join(d, "box.obj"), box_v, box_f)
    eye, cam = H.camera_orbit(0.8, 8.0, 10.0, center=(0.26, -1.28, 0.0))
    objs = [  # file, transform, emissive, brdf, reflex, refract, rate, albedo, index, normalise  (PathTrace.cu:1500-1524)
        ("statue.obj", H.transform_matrix((-90, 0, 0), (0, -0.52, 0.5), (0.3, 0.3, 0.3)), "0 0 0", "0.02 0.02 0.02", 1, 1, "0.1 0.1 0.1", "0.3 0.3 0.3", "2.66", 1),
        ("light.obj", H.transform_matrix((0, 90, 90), (-0.2, 1.2, 1.0), (1.5, 0.5, 1.5)), "1000 1000 1000", "0.3 0.3 0.3", 0, 0, "0.8 0.8 0.8", "0.8 0.8 0.8", "1", 1),
        ("box.obj", H.transform_matrix((0, 0, 0), (0, -0.5625, 0), (12, 0.125, 12)), "0 0 0", "0.3 0.3 0.3", 1, 0, "0.7 0.7 0.7", "0.3 0.3 0.3", "1.1", 1),
        ("ball.obj", H.transform_matrix((0, 0, 0), (0.25, -0.35, 0.45), (0.08, 0.08, 0.08)), "0 0 0", "0.05 0.05 0.05", 1, 2, "0.9 0.95 0.9", "0.3 0.3 0.3", "1.5", 0),
    ]
    lines = [" ".join(f"{x:.9g}" for x in eye)] + _mat4(cam) + [str(len(objs))]
    for f, t, em, brdf, reflex, refract, rate, albedo, idx, norm in objs:
        lines += [f] + _mat4(t) + [em, brdf, str(reflex), str(refract), rate, albedo, idx, str(norm)]
    args = os.path.join(d, "render_args.txt")
    open(args, "w").write("\n".join(lines) + "\n")
    # an RGBE environment: vertical gradient + a bright patch (values beyond the sampler's clamp of 10), run-length encoded
    hh, ww = 32, 64
    yy = np.linspace(0, 1, hh, dtype=np.float32)[:, None, None]
    env = np.broadcast_to(np.float32([0.9, 1.1, 1.6]) * (1.2 - yy), (hh, ww, 3)).copy()
    env[4:8, 20:26] = [40.0, 36.0, 30.0]
    hdr = os.path.join(d, "background.hdr")
    _write_hdr(hdr, env, rle=True)

    size, spp = (96, 64), 6
    st_o, _ = _run(ORACLE_LIB, args, hdr, os.path.join(d, "o.pfm"), size, spp)
    st_h, log = _run(HIP_LIB, args, hdr, os.path.join(d, "h.pfm"), size, spp, ["--reference-walk"])
    assert "hip-gfx950" in log
    keys = ("rays_primary", "rays_secondary", "nodes_visited", "tris_tested", "shaded_hits", "samples")
    assert {k: st_h[k] for k in keys} == {k: st_o[k] for k in keys}                      # every decision identical
    # the CLI's default walk (early exits, jade_rt.h): the same file byte for byte, the same rays, fewer node records and tests
    st_e, _ = _run(HIP_LIB, args, hdr, os.path.join(d, "e.pfm"), size, spp)
    assert open(os.path.join(d, "e.pfm"), "rb").read() == open(os.path.join(d, "h.pfm"), "rb").read()
    assert {k: st_e[k] for k in keys if k not in ("nodes_visited", "tris_tested")} == {k: st_h[k] for k in keys if k not in ("nodes_visited", "tris_tested")}
    assert st_e["nodes_visited"] < st_h["nodes_visited"] and st_e["tris_tested"] < st_h["tris_tested"]
    assert st_h["samples"] == size[0] * size[1] * spp and st_h["rays_secondary"] > st_h["samples"] // 4
    a, b = _read_pfm(os.path.join(d, "h.pfm")), _read_pfm(os.path.join(d, "o.pfm"))
    assert a.shape == (size[1], size[0], 3) and rel_l2(a, b) <= 1e-4
    # and the reference's own output format: 24-bit BMP, at most one code value apart
    _run(ORACLE_LIB, args, hdr, os.path.join(d, "o.bmp"), size, spp)
    _run(HIP_LIB, args, hdr, os.path.join(d, "h.bmp"), size, spp)
    ra, rb = open(os.path.join(d, "h.bmp"), "rb").read(), open(os.path.join(d, "o.bmp"), "rb").read()
    assert ra[:54] == rb[:54] and len(ra) == 54 + size[0] * size[1] * 3
    diff = np.abs(np.frombuffer(ra[54:], np.uint8).astype(int) - np.frombuffer(rb[54:], np.uint8).astype(int))
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-3
