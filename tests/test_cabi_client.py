"""tests/cpp/cabi_client.c: a plain-C program that drives create -> textures/materials -> staging + commit -> raycast -> render ->
presentation through include/voxel_hip.h alone -- the call sequence of integration/rust/svo_hip.rs, without Python or the C++
mirror in between. It compiles as C11 against the header (so the header is valid C) and links the product library."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BUILD = ROOT / "tests" / "_build"


def build_client():
    BUILD.mkdir(exist_ok=True)
    exe = BUILD / "cabi_client"
    lib = ROOT / "voxel-rs_amd" / "lib"
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", f"-I{ROOT}/include", str(ROOT / "tests" / "cpp" / "cabi_client.c"), f"-L{lib}", "-lvoxelhip",
           f"-Wl,-rpath,{lib}", "-lm", "-o", str(exe)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_c_client_builds_and_refuses_without_a_gpu():
    import torch

    exe = build_client()
    if torch.cuda.device_count() > 0:
        return  # the GPU test below runs it for real
    r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 2 and "no HIP device" in r.stdout, r.stdout  # no CPU path: vx_create says VX_ERR_NO_DEVICE


@pytest.mark.gpu
def test_c_client_end_to_end():
    r = subprocess.run([str(build_client())], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "cabi_client: ok" in r.stdout, r.stdout
