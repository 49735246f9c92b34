"""Builds the in-tree native libraries with make (hipcc cross-compiles gfx950 without a GPU)."""
import os
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
ROOT = PKG.parent
# VX_LIB_DIR (measurements: A/B of two builds of the libraries in one gpurun call) overrides the in-tree lib directory
LIB = Path(os.environ["VX_LIB_DIR"]).resolve() if os.environ.get("VX_LIB_DIR") else PKG / "lib"


def _run(cmd, cwd):
    env = dict(os.environ)
    env.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    r = subprocess.run(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)} failed in {cwd}:\n{r.stdout}")
    return r.stdout


def build_all(jobs=4):
    """Compiles libvoxelhip.so, libvoxelhost.so and the KAT binary. (The CPU oracle is test infrastructure and is built by its own
    callers: __graft_entry__.build() and oracle/oracle.py.)"""
    return _run(["make", f"-j{jobs}", "-C", str(PKG), "all"], ROOT)


def lib_path(name):
    p = LIB / name
    if not p.exists():
        raise FileNotFoundError(f"{p} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (or make -C voxel-rs_amd)")
    return p


def share_hip_runtime_with_torch():
    """The PyTorch wheel bundles its own libamdhip64 (same SONAME as /opt/rocm's). Whichever copy is loaded first serves the
    whole process; loading ours first and torch's later puts two HIP runtimes in one process and torch then reports
    "No HIP GPUs". When torch is installed, import it before any of our libraries so that both share one runtime."""
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
