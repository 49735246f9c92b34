"""Synthetic scene generator: determinism, integer heightfield reproducible from Python, format-independent content."""
import numpy as np
import pytest

from helpers import SVO_TYPES, orc, vra
from voxel_rs_amd import host, scenes


def py_hash32(seed, o, i, j):
    M = 0xFFFFFFFF
    h = (seed ^ (o * 0x9E3779B1)) & M
    h = ((h ^ i) * 0x85EBCA6B) & M
    h ^= h >> 13
    h = ((h ^ j) * 0xC2B2AE35) & M
    h ^= h >> 16
    h = (h * 0x27D4EB2F) & M
    h ^= h >> 15
    return h


def py_height(depth, seed, x, z):
    n = 1 << depth
    height = 1
    for o in range(5):
        lam = max(n >> (2 + o), 1)
        amp = (n // 8) >> o
        i, j, fx, fz = x // lam, z // lam, x % lam, z % lam
        v = [py_hash32(seed, o, i + a, j + b) & 0xFFFF for b in (0, 1) for a in (0, 1)]
        a = v[0] * (lam - fx) + v[1] * fx
        b = v[2] * (lam - fx) + v[3] * fx
        height += (((a * (lam - fz) + b * fz) // (lam * lam)) * amp) >> 16
    return min(max(height, 1), max(n // 4, 1))


def test_heightfield_is_reproducible_from_python():
    rng = np.random.default_rng(0)
    for depth in (6, 9, 12):
        for _ in range(50):
            x, z = (int(v) for v in rng.integers(0, 1 << depth, size=2))
            assert host.scene_height(depth, 0x5EED0001, x, z) == py_height(depth, 0x5EED0001, x, z)


@pytest.mark.parametrize("fmt", ["esvo", "csvo"])
def test_scene_build_is_deterministic_and_thread_independent(fmt):
    frames = []
    for threads in (1, 4):
        w = vra.World(SVO_TYPES[fmt])
        st = w.build_heightfield(7, threads=threads)
        frames.append((st, w.frame().tobytes()))
        assert w.depth == 7
    assert frames[0] == frames[1]
    st = frames[0][0]
    assert st["chunks"] > 0 and st["leaves"] >= (1 << 7) ** 2  # at least one voxel per column


def test_both_formats_describe_the_same_voxels():
    """ESVO and CSVO serialisations of one scene give identical oracle hits for the same rays."""
    res = {}
    for fmt in ("esvo", "csvo"):
        w = vra.World(SVO_TYPES[fmt])
        st = w.build_heightfield(7, threads=2)
        sc = orc.OracleScene(SVO_TYPES[fmt], w.frame(), scenes.synthetic_materials().view(orc.MATERIAL_DTYPE), scenes.synthetic_textures(), 6)
        u = scenes.bench_camera(7, st["h_max"], 96, 64)
        img, hits = sc.render(orc.Uniforms.from_buffer_copy(bytes(u)), 96, 64)
        res[fmt] = (img, hits)
    assert res["esvo"][1].tobytes() == res["csvo"][1].tobytes()
    assert np.array_equal(np.nan_to_num(res["esvo"][0]), np.nan_to_num(res["csvo"][0]))
    assert (res["esvo"][1]["flags"] & 1).mean() > 0.2


def test_synthetic_registry_shape():
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    assert tex.shape == (25, 64, 64, 4) and mats.size == 13  # content.rs:23-60
    assert (tex[scenes.TEXTURE_NAMES.index("glass"), :, :, 3] == 0).any()
    assert mats[1]["tex_top"] == scenes.TEXTURE_NAMES.index("grass_top") and mats[1]["tex_bottom"] == scenes.TEXTURE_NAMES.index("dirt")
    assert mats[5]["tex_side_normal"] == -1 and mats[3]["tex_side_normal"] == scenes.TEXTURE_NAMES.index("stone_normal")
