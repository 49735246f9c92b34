"""Synthetic inputs for benchmarks and at-scale parity runs: a procedural block registry shaped like the
reference's (13 materials, 25 texture layers of 64x64, src/gamelogic/content.rs:20-62), the §8d camera, and the
`RenderParams -> uniforms` step of `graphics::Svo::render` (src/graphics/svo.rs:196-215).

Textures are generated (seeded integer hash), not copied: /root/reference does not exist on the GPU box.
"""
import math

import numpy as np

from .hip import MATERIAL_DTYPE, make_uniforms

# texture table in the order content.rs registers it (:23-47); names ending in _normal are normal maps
TEXTURE_NAMES = [
    "dirt", "dirt_normal", "grass_side", "grass_side_normal", "grass_top", "grass_top_normal", "stone", "stone_normal", "stone_bricks",
    "stone_bricks_normal", "glass", "gravel", "gravel_normal", "sand", "sand_normal", "water", "oak_log", "oak_log_normal", "oak_log_top",
    "oak_log_top_normal", "oak_leaves", "oak_planks", "oak_planks_normal", "cobblestone", "cobblestone_normal",
]
_BASE_COLOR = {
    "dirt": (134, 96, 67), "grass_side": (120, 110, 70), "grass_top": (96, 160, 64), "stone": (125, 125, 125), "stone_bricks": (110, 110, 115),
    "glass": (200, 230, 240), "gravel": (130, 125, 120), "sand": (218, 210, 158), "water": (50, 90, 200), "oak_log": (102, 81, 50),
    "oak_log_top": (150, 120, 80), "oak_leaves": (60, 130, 50), "oak_planks": (160, 130, 80), "cobblestone": (120, 120, 120),
}


def _hash32(a):
    a = (a ^ 61) ^ (a >> 16)
    a = (a + (a << 3)) & 0xFFFFFFFF
    a ^= a >> 4
    a = (a * 0x27D4EB2D) & 0xFFFFFFFF
    a ^= a >> 15
    return a


def synthetic_textures(seed=1, size=64):
    """uint8 [25][size][size][4], row 0 = bottom. glass / oak_leaves carry alpha = 0 texels (translucency path)."""
    layers = np.zeros((len(TEXTURE_NAMES), size, size, 4), dtype=np.uint8)
    ys, xs = np.mgrid[0:size, 0:size]
    for li, name in enumerate(TEXTURE_NAMES):
        h = np.vectorize(_hash32)((xs * 73856093 ^ ys * 19349663 ^ (li + 1) * 83492791 ^ seed).astype(np.uint64) & 0xFFFFFFFF).astype(np.uint32)
        n = (h & 0xFF).astype(np.int32)
        if name.endswith("_normal"):
            # tangent-space normals around (0,0,1), encoded 0..255; .xzy swizzle happens in the shader (world.glsl:60)
            layers[li, :, :, 0] = np.clip(128 + ((h >> 8) & 0x3F).astype(np.int32) - 32, 0, 255)
            layers[li, :, :, 1] = np.clip(128 + ((h >> 16) & 0x3F).astype(np.int32) - 32, 0, 255)
            layers[li, :, :, 2] = 235
            layers[li, :, :, 3] = 255
            continue
        r, g, b = _BASE_COLOR[name]
        jitter = (n - 128) // 6
        layers[li, :, :, 0] = np.clip(r + jitter, 0, 255)
        layers[li, :, :, 1] = np.clip(g + jitter, 0, 255)
        layers[li, :, :, 2] = np.clip(b + jitter, 0, 255)
        layers[li, :, :, 3] = 255
        if name == "glass":
            inner = (xs > 3) & (xs < size - 4) & (ys > 3) & (ys < size - 4)
            layers[li, :, :, 3] = np.where(inner, 0, 255)  # frame opaque, pane fully translucent
        if name == "oak_leaves":
            layers[li, :, :, 3] = np.where((h >> 24) & 3, 255, 0)  # a quarter of the texels are holes
        if name == "grass_side":
            top = ys >= size - 12
            layers[li, :, :, 1] = np.where(top, np.clip(150 + jitter, 0, 255), layers[li, :, :, 1])
    return layers


def asset_textures(directory, seed=1, size=64):
    """The 25-layer table with the 64x64 PNGs found in `directory` (the caller's: bench.py and the tests pass the repository's
    fixtures -- dirt, grass_side, grass_top, stone and their normal maps, byte-identical copies of assets/textures/*.png, the eight
    the reference's render test loads, src/graphics/svo.rs:347-360) and the procedural stand-ins for the rest. Flipped vertically like TextureArrayBuilder does (texture_array.rs:92,126): row 0 = bottom. The benchmark's
    terrain only uses grass, dirt and stone, i.e. only real textures."""
    from pathlib import Path

    from PIL import Image

    directory = Path(directory)
    if not directory.is_dir():
        raise FileNotFoundError(f"texture directory {directory} does not exist")
    layers = synthetic_textures(seed, size)
    for li, name in enumerate(TEXTURE_NAMES):
        f = directory / (name[:-7] + "_n.png" if name.endswith("_normal") else name + ".png")
        if f.exists():
            im = np.asarray(Image.open(f).convert("RGBA"), dtype=np.uint8)
            if im.shape == (size, size, 4):
                layers[li] = im[::-1]
    return layers


def synthetic_materials():
    """13 rows indexed by BlockId with the specular parameters and face/texture wiring of content.rs:48-60."""
    idx = {n: i for i, n in enumerate(TEXTURE_NAMES)}

    def tex(name):
        return idx[name] if name else -1

    def nrm(name):
        return idx.get(name + "_normal", -1) if name else -1

    rows = [  # (pow, strength, top, side, bottom, with_normals)
        (0.0, 0.0, None, None, None, False),                      # AIR
        (14.0, 0.4, "grass_top", "grass_side", "dirt", True),     # GRASS
        (14.0, 0.4, "dirt", "dirt", "dirt", True),                # DIRT
        (70.0, 0.4, "stone", "stone", "stone", True),             # STONE
        (70.0, 0.4, "stone_bricks", "stone_bricks", "stone_bricks", True),
        (70.0, 0.4, "glass", "glass", "glass", False),
        (70.0, 0.4, "gravel", "gravel", "gravel", True),
        (70.0, 0.4, "sand", "sand", "sand", True),
        (70.0, 0.4, "water", "water", "water", False),
        (70.0, 0.4, "oak_log_top", "oak_log", "oak_log_top", True),
        (70.0, 0.4, "oak_leaves", "oak_leaves", "oak_leaves", False),
        (70.0, 0.4, "oak_planks", "oak_planks", "oak_planks", True),
        (70.0, 0.4, "cobblestone", "cobblestone", "cobblestone", True),
    ]
    m = np.zeros(len(rows), dtype=MATERIAL_DTYPE)
    for i, (p, s, top, side, bottom, normals) in enumerate(rows):
        m[i] = (p, s, tex(top), tex(side), tex(bottom), nrm(top) if normals else -1, nrm(side) if normals else -1, nrm(bottom) if normals else -1)
    return m


def _normalize(v):
    v = np.asarray(v, dtype=np.float32)
    return (v / np.sqrt(np.float32(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]), dtype=np.float32)).astype(np.float32)


def view_matrix(eye, fwd, up):
    """u_view = look_to_rh(eye, fwd, up)^-1 (svo.rs:197; cgmath 0.18): columns [s, u, -f, eye], column-major."""
    f = _normalize(fwd)
    s = _normalize(np.cross(f, np.asarray(up, dtype=np.float32)).astype(np.float32))
    u = np.cross(s, f).astype(np.float32)
    m = np.zeros(16, dtype=np.float32)
    m[0:3], m[4:7], m[8:11], m[12:15], m[15] = s, u, -f, np.asarray(eye, dtype=np.float32), 1.0
    return m


def render_params_to_uniforms(cam_pos, cam_fwd, cam_up, fov_y_rad, aspect, ambient=0.3, light_dir=(-1.0, -1.0, -1.0), render_shadows=True,
                              shadow_distance=500.0, selected_voxel=None):
    """RenderParams (svo.rs:85-106) -> the uniform block Svo::render uploads (svo.rs:201-215)."""
    return make_uniforms(view_matrix(cam_pos, cam_fwd, cam_up), fov_y_rad, aspect, ambient, _normalize(light_dir), cam_pos, render_shadows,
                         shadow_distance, selected_voxel)


def bench_camera(depth, h_max, width, height, shadow_distance=500.0, render_shadows=True):
    """The §8d camera: eye above the terrain centre, looking along (0.6,-0.35,0.7), fovy 72 degrees
    (src/main.rs:97), sun (-1,-1,-1), ambient 0.3, shadow distance 500 (src/gamelogic/world.rs:105-108)."""
    n = float(1 << depth)
    eye = (0.5 * n, h_max + 0.05 * n, 0.5 * n)
    return render_params_to_uniforms(eye, (0.6, -0.35, 0.7), (0.0, 1.0, 0.0), math.radians(72.0), width / height, 0.3, (-1.0, -1.0, -1.0),
                                     render_shadows, shadow_distance)
