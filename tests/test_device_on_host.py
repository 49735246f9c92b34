"""The DEVICE traversal code (voxel-rs_amd/csrc/hip/vx_device.hpp), compiled for the host by a test-only harness
(tests/cpp/device_on_host.cpp: shims for the HIP built-ins, nothing from the product links against it), stepped against
the oracle on seeded picker rays. It catches logic slips in the kernels' shared header on a machine without a GPU; the GPU
parity tests remain the ones that count. Results must be identical byte for byte (48-byte PickerResult records)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

from helpers import ROOT, golden_materials, golden_textures, oracle_scene, orc, vra

BUILD = Path(ROOT) / "tests" / "_build"
NORMALS = [[-1, 0, 0], [1, 0, 0], [0, -1, 0], [0, 1, 0], [0, 0, -1], [0, 0, 1]]


@pytest.fixture(scope="module")
def devhost():
    BUILD.mkdir(exist_ok=True)
    so = BUILD / "libdevice_on_host.so"
    src = Path(ROOT) / "tests" / "cpp" / "device_on_host.cpp"
    hdr = Path(ROOT) / "voxel-rs_amd" / "csrc" / "hip" / "vx_device.hpp"
    if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        cmd = ["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", f"-I{ROOT}/include",
               f"-I{ROOT}/voxel-rs_amd/csrc/hip", str(src), "-o", str(so)]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout
    return C.CDLL(str(so))


def expected(scene, tasks, cast_translucent):
    exp = np.zeros(len(tasks), dtype=orc.PICKER_RESULT_DTYPE)
    for i, t in enumerate(tasks):
        r, _, _ = scene.intersect(t["pos"], t["dir"], float(t["max_dst"]), bool(cast_translucent))
        if r.t > 0:
            exp[i]["dst"] = r.t
            exp[i]["inside_voxel"] = r.inside_voxel
            exp[i]["pos"] = list(r.pos)
            exp[i]["normal"] = NORMALS[r.face_id]
        else:
            exp[i]["dst"] = -1
    return exp


def run(lib, golden, fmt, world, tasks, cast_translucent):
    frame = world.frame(pad_words=0)
    tex, _ = golden_textures(golden)
    mats = golden_materials(golden)
    out = np.zeros(len(tasks), dtype=orc.PICKER_RESULT_DTYPE)
    level_offset = (C.c_uint32 * 16)(0)
    lib.devhost_picker(1 if fmt == "esvo" else 2, frame.ctypes.data_as(C.c_void_p), C.c_uint64(frame.size * 4), mats.ctypes.data_as(C.c_void_p),
                       mats.size, tex.ctypes.data_as(C.c_void_p), 4, 4, 4, 1, level_offset, tasks.ctypes.data_as(C.c_void_p), len(tasks),
                       out.ctypes.data_as(C.c_void_p), cast_translucent)
    return out


def random_tasks(rng, n, lo, hi):
    tasks = np.zeros(n, dtype=orc.PICKER_TASK_DTYPE)
    tasks["pos"] = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::13] = np.eye(3)[rng.integers(0, 3, size=len(d[::13]))]  # axis-parallel rays: the epsilon clamp of svo.esvo.glsl:85-89
    tasks["dir"] = d.astype(np.float32)
    tasks["max_dst"] = np.where(rng.random(n) < 0.3, rng.uniform(1, 40, size=n), -1).astype(np.float32)
    return tasks


@pytest.mark.parametrize("cast_translucent", [0, 1])
@pytest.mark.parametrize("fmt", ["esvo", "csvo"])
@pytest.mark.parametrize("seed,svo_pos,n_blocks", [(2, (1, 0, 1), 600), (3, (3, 2, 1), 6000)])
def test_device_traversal_on_host_matches_oracle(devhost, golden, fmt, seed, svo_pos, n_blocks, cast_translucent):
    rng = np.random.default_rng(seed)
    pts = rng.integers(0, 32, size=(n_blocks, 3))
    blocks = [[int(x), int(y), int(z), int(rng.choice([1, 2, 3, 4]))] for x, y, z in pts]
    scene, world = oracle_scene(golden, fmt, svo_pos, blocks)
    base = np.asarray(svo_pos, dtype=np.float32) * 32
    tasks = random_tasks(rng, 1500, -8, 40)
    tasks["pos"] += base  # origins outside, on the border of and inside the chunk (and inside voxels)
    got, exp = run(devhost, golden, fmt, world, tasks, cast_translucent), expected(scene, tasks, cast_translucent)
    assert (exp["dst"] > 0).sum() > 100
    assert (exp["inside_voxel"] != 0).sum() > 0
    bad = [i for i in range(len(tasks)) if got[i].tobytes() != exp[i].tobytes()]
    assert not bad, (len(bad), bad[:5])


def test_device_traversal_on_host_heightfield_csvo(devhost, golden):
    world = vra.World(2)
    world.build_heightfield(8, threads=4)
    tex, mips = golden_textures(golden)
    scene = orc.OracleScene(2, world.frame(), golden_materials(golden), tex, mips)
    rng = np.random.default_rng(5)
    tasks = random_tasks(rng, 1500, 0, 256)
    tasks["pos"] *= np.float32([1, 0.5, 1])
    tasks["max_dst"] = -1
    got, exp = run(devhost, golden, "csvo", world, tasks, 1), expected(scene, tasks, 1)
    assert (exp["dst"] > 0).sum() > 300
    bad = [i for i in range(len(tasks)) if got[i].tobytes() != exp[i].tobytes()]
    assert not bad, (len(bad), bad[:5])
