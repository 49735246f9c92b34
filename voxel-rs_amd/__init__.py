"""voxel-rs_amd: MI355X-native SVO raycaster behind voxel-rs's `graphics::Svo` render/raycast surface.

The product is `lib/libvoxelhip.so` (C-ABI in include/voxel_hip.h, HIP kernels for gfx950) plus
`lib/libvoxelhost.so` (C++ mirror of the reference's Rust host code). This Python package is only the
harness-side binding (ctypes) used by tests, bench.py and __graft_entry__.py.

The directory name is not a Python identifier; load it with `load_package()` from `_pkg.py` at the repo root
or via `importlib` under the alias `voxel_rs_amd` (tests/conftest.py does that).
"""
from . import build as build  # noqa: F401
from .host import Chunk, World, SVO_CSVO, SVO_ESVO  # noqa: F401
