"""jaderaytracerendering_amd — MI355X-native jade/BSSRDF path tracer.

Layout:
  csrc/     hand-written HIP for gfx950 + the jade_rt.h C ABI  -> lib/libjade_hip.so
  host/     the repo's own C++ host side (scene loading, SAH BVH, camera,
            image writers, CLI)                                 -> lib/libjade_host.so
  host.py, backend.py   ctypes mirrors of the two C interfaces

The product path is `backend.hip()`; it raises when the HIP library is
missing.  The CPU oracle under oracle/ is test infrastructure and is never
imported from here.
"""
from . import _abi, backend, host  # noqa: F401
from .backend import Backend, JadeError, hip, make_params, params_from_config  # noqa: F401
from .host import HostScene, SceneBuilder, build_config  # noqa: F401

__all__ = ["Backend", "JadeError", "hip", "make_params", "params_from_config", "HostScene", "SceneBuilder",
           "build_config", "backend", "host"]
