"""Loads the in-tree package directory `voxel-rs_amd/` (not a Python identifier) as module `voxel_rs_amd`."""
import importlib.util
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent


def load_package():
    if "voxel_rs_amd" in sys.modules:
        return sys.modules["voxel_rs_amd"]
    pkg_dir = ROOT / "voxel-rs_amd"
    spec = importlib.util.spec_from_file_location("voxel_rs_amd", pkg_dir / "__init__.py", submodule_search_locations=[str(pkg_dir)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["voxel_rs_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def csrc_hash():
    """sha256[:16] over the names and contents of the library's device and runtime sources (voxel-rs_amd/csrc/hip/*): what a stored counter file
    (profiles/roundN/traffic.json) was measured on, checked by bench.py without git (the GPU box's snapshot has no history)."""
    import hashlib

    h = hashlib.sha256()
    for f in sorted((ROOT / "voxel-rs_amd" / "csrc" / "hip").iterdir()):
        if f.is_file():
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]
