"""Shared test helpers: build the reference's test worlds / textures / materials from the golden fixture."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from _pkg import load_package  # noqa: E402

vra = load_package()
from oracle import oracle as orc  # noqa: E402  (the checker; tests are allowed to use it)

SVO_TYPES = {"esvo": vra.SVO_ESVO, "csvo": vra.SVO_CSVO}


def build_world(svo_type, svo_pos, blocks, compact=True):
    """create_test_world of src/graphics/svo_shader_tests.rs:78-115."""
    chunk = vra.Chunk(0, 0, 0, lod=5)
    chunk.apply_blocks(blocks)
    if compact:
        chunk.compact()
    world = vra.World(svo_type)
    world.set_chunk(tuple(svo_pos), chunk, True)
    world.serialize()
    return world


def golden_textures(golden):
    """4 layers of 4x4 RGBA8, flipped vertically like TextureArrayBuilder::add_rgba8 (texture_array.rs:67-69)."""
    t = golden["textures"]
    layers = []
    for name in t["order"]:
        rows = np.array(t["rgba8_rows_top_to_bottom"][name], dtype=np.uint8).reshape(t["height"], t["width"], 4)
        layers.append(rows[::-1])  # row 0 = bottom
    return np.ascontiguousarray(np.stack(layers)), t["mip_levels"]


def golden_materials(golden):
    mats = np.zeros(len(golden["materials"]), dtype=orc.MATERIAL_DTYPE)
    for i, m in enumerate(golden["materials"]):
        for k, v in m.items():
            mats[i][k] = v
    return mats


def oracle_scene(golden, fmt, svo_pos, blocks, compact=True):
    world = build_world(SVO_TYPES[fmt], svo_pos, blocks, compact)
    tex, mips = golden_textures(golden)
    return orc.OracleScene(SVO_TYPES[fmt], world.frame(), golden_materials(golden), tex, mips), world
