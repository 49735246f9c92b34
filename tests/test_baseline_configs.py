"""BASELINE.json's configurations at their full sizes, HIP path against the oracle:
C1 256x256 rays against one 32^3 chunk through the picker path; C2 1920x1080 primary rays only on a depth-10 SVO;
C3 1920x1080 primary + shadow, textured and normal-mapped, on the depth-12 SVO the benchmark uses. Every hit record is
compared exactly (t, position, uv, value, face, flags, shadow distance, step count); colours to 5e-6."""
import math
from pathlib import Path

import numpy as np
import pytest

from helpers import orc, vra
from voxel_rs_amd import scenes

pytestmark = pytest.mark.gpu
FMTS = {"esvo": vra.SVO_ESVO, "csvo": vra.SVO_CSVO}
SEED = 0x5EED0001


@pytest.fixture(scope="module")
def hip():
    from voxel_rs_amd import hip as h

    return h


def hash32(seed, o, i, j):
    from voxel_rs_amd import host

    return int(host.lib().vxh_scene_hash32(seed, o, i, j))


@pytest.mark.parametrize("fmt", FMTS)
def test_c1_picker_rays_against_one_chunk(hip, fmt):
    """One chunk at SVO position (0,0,0); voxel (x,y,z) solid iff y <= 8 + (hash32(seed,x,z) & 7); 256x256 perspective rays
    from eye (16,24,-24) towards (16,8,16), each a PickerTask with max_dst 100, cast_translucent = false (picker.glsl)."""
    chunk = vra.Chunk(0, 0, 0, 5)
    for x in range(32):
        for z in range(32):
            top = 8 + (hash32(SEED, 0, x, z) & 7)
            for y in range(top + 1):
                chunk.set_block(x, y, z, 1 if y == top else (2 if y + 3 >= top else 3))
    chunk.compact()
    world = vra.World(FMTS[fmt])
    world.set_chunk((0, 0, 0), chunk)
    world.serialize()
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    scene = orc.OracleScene(FMTS[fmt], world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    svo = hip.Svo(FMTS[fmt], 4 << 20)
    svo.set_materials(mats)
    svo.set_textures(tex, 6)
    svo.update(world)

    n = 256
    eye = np.float32([16, 24, -24])
    fwd = np.float32([16, 8, 16]) - eye
    u = scenes.render_params_to_uniforms(tuple(eye), tuple(fwd), (0.0, 1.0, 0.0), math.radians(72.0), 1.0, 0.3, (-1.0, -1.0, -1.0), False, 500.0)
    ou = orc.Uniforms.from_buffer_copy(bytes(u))
    tasks = np.zeros(n * n, dtype=orc.PICKER_TASK_DTYPE)
    import ctypes as C

    ro, rd = (C.c_float * 3)(), (C.c_float * 3)()
    for y in range(n):
        for x in range(n):
            orc.lib().or_primary_ray(C.byref(ou), n, n, x, y, C.byref(ro), C.byref(rd))
            t = tasks[y * n + x]
            t["max_dst"], t["pos"], t["dir"] = 100.0, list(ro), list(rd)
    got = svo.raycast(tasks)  # one call: vx_raycast has no 100-task cap (svo_picker.rs:5)
    exp = scene.picker(tasks, threads=8)
    assert got.tobytes() == exp.tobytes()
    assert (exp["dst"] > 0).mean() > 0.3


TEXTURE_DIR = Path(__file__).resolve().parent / "golden" / "textures"

# (config, textures, shadow distance): C3 with the reference's textures and an unlimited shadow distance is the frame bench.py times
# (`--textures assets`, every primary hit casts its shadow ray); 500 blocks is the game's own cut-off (src/gamelogic/world.rs:105-108),
# bench.py's second line
FRAMES = [("C2", "synthetic", 3.0e38), ("C3", "synthetic", 3.0e38), ("C3", "assets", 3.0e38), ("C3", "assets", 500.0)]


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("config,textures,shadow_distance", FRAMES, ids=[f"{c}-{t}-{'inf' if d > 1e30 else int(d)}" for c, t, d in FRAMES])
def test_full_size_frames(hip, fmt, config, textures, shadow_distance):
    depth, shadows = (10, False) if config == "C2" else (12, True)
    world = vra.World(FMTS[fmt])
    st = world.build_heightfield(depth)
    tex = scenes.asset_textures(TEXTURE_DIR) if textures == "assets" else scenes.synthetic_textures()
    mats = scenes.synthetic_materials()
    scene = orc.OracleScene(FMTS[fmt], world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    svo = hip.Svo(FMTS[fmt], world.size_in_bytes + (16 << 20))
    svo.set_materials(mats)
    svo.set_textures(tex, 6)
    svo.update(world)
    w, h = 1920, 1080
    u = scenes.bench_camera(depth, st["h_max"], w, h, shadow_distance=shadow_distance, render_shadows=shadows)
    img, hits = svo.render(u, w, h, want_hits=True)
    cimg, chits = scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), w, h)
    assert hits.tobytes() == chits.tobytes()
    assert np.array_equal(np.isnan(img), np.isnan(cimg))
    assert np.nanmax(np.abs(img - cimg)) <= 5e-6
    primary_hits = int((chits["flags"] & 1).sum())
    shadow_rays = int(((chits["flags"] >> 1) & 1).sum())
    assert primary_hits > 0.4 * w * h
    if not shadows:
        assert shadow_rays == 0
    elif shadow_distance > 1e30:
        assert shadow_rays == primary_hits
    else:
        # world.glsl:80: only hits nearer than the cut-off cast one (from this altitude: few or none -- the frame is then primary rays only)
        assert shadow_rays == int((((chits["flags"] & 1) != 0) & (chits["t"] < shadow_distance)).sum()) < primary_hits
    # The image-only kernel (no hit records) with the library's default of two frames in flight -- exactly what bench.py times --
    # against the ORACLE's frame (not only against the hit-record kernel's), and identical to the hit-record kernel's pixels.
    import torch

    svo.set_frames_in_flight(2)  # (bench.py's call; also the library's default)
    a = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    b = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()  # torch zero-fills on ITS stream; the renderer's streams do not wait for it
    for _ in range(3):
        svo.render_device(u, w, h, a.data_ptr())
        svo.render_device(u, w, h, b.data_ptr())
    svo.sync()
    for t in (a, b):
        got = t.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(cimg)) and np.nanmax(np.abs(got - cimg)) <= 5e-6
        assert np.array_equal(np.isnan(got), np.isnan(img)) and np.nanmax(np.abs(got - img)) == 0.0


@pytest.mark.parametrize("fmt", FMTS)
def test_views_of_the_benchmarks_moving_camera(hip, fmt):
    """The frames bench.py TIMES: its own twenty views (bench.moving_uniforms: 1920x1080, depth 12, the reference's textures, every primary hit casts
    its shadow ray), rendered as it renders them -- image-only, device resident, two frames in flight, one after the other along the path -- and the
    first, the eighth and the last of them against the oracle's frames (colour within 5e-6; the render with hit records of the same views: exact
    records, and the image-only frame's pixels byte for byte)."""
    import sys

    import torch

    root = Path(__file__).resolve().parent.parent
    if str(root) not in sys.path:
        sys.path.insert(0, str(root))
    import bench

    depth, w, h, steps = 12, 1920, 1080, 20
    world = vra.World(FMTS[fmt])
    st = world.build_heightfield(depth)
    tex, mats = scenes.asset_textures(TEXTURE_DIR), scenes.synthetic_materials()
    scene = orc.OracleScene(FMTS[fmt], world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    svo = hip.Svo(FMTS[fmt], world.size_in_bytes + (16 << 20))
    svo.set_materials(mats)
    svo.set_textures(tex, 6)
    svo.update(world)
    svo.set_frames_in_flight(2)
    path = [bench.moving_uniforms(scenes, depth, st["h_max"], w, h, i) for i in range(steps)]
    frames = [torch.zeros((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(steps)]
    torch.cuda.synchronize()
    for _ in range(2):  # (twice along the path, like consecutive timed blocks)
        for i, u in enumerate(path):
            svo.render_device(u, w, h, frames[i].data_ptr())
    svo.sync()
    for i in (0, 7, 19):
        got = frames[i].cpu().numpy()
        cimg, chits = scene.render(orc.Uniforms.from_buffer_copy(bytes(path[i])), w, h)
        assert np.array_equal(np.isnan(got), np.isnan(cimg)), f"view {i}"
        assert np.nanmax(np.abs(got - cimg)) <= 5e-6, f"view {i}"
        img, hits = svo.render(path[i], w, h, want_hits=True)
        assert hits.tobytes() == chits.tobytes(), f"view {i}"
        assert np.array_equal(np.isnan(got), np.isnan(img)) and np.nanmax(np.abs(got - img)) == 0.0, f"view {i}"
        assert int((chits["flags"] & 1).sum()) > 0.3 * w * h


@pytest.mark.parametrize("fmt", FMTS)
def test_c5_supersampled_frame(hip, fmt):
    """2x2 ordered-grid supersampling: a (2w x 2h) render box-filtered down, against the oracle's (2w x 2h) frame filtered
    the same way in numpy (same association order: (a + b) + (c + d), then * 0.25)."""
    import torch

    depth = 9
    world = vra.World(FMTS[fmt])
    st = world.build_heightfield(depth, threads=4)
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    scene = orc.OracleScene(FMTS[fmt], world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    svo = hip.Svo(FMTS[fmt], world.size_in_bytes + (1 << 20))
    svo.set_materials(mats)
    svo.set_textures(tex, 6)
    svo.update(world)
    w, h = 480, 270
    u = scenes.bench_camera(depth, st["h_max"], 2 * w, 2 * h, shadow_distance=3.0e38, render_shadows=True)
    big = torch.zeros((2 * h, 2 * w, 4), dtype=torch.float32, device="cuda")
    small = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    svo.render_device(u, 2 * w, 2 * h, big.data_ptr())
    stream = torch.cuda.current_stream().cuda_stream
    svo.stream_wait_render(stream)
    svo.resolve_2x2(big.data_ptr(), w, h, small.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    cimg, _ = scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), 2 * w, 2 * h, want_hits=False)
    ref = ((cimg[0::2, 0::2] + cimg[0::2, 1::2]) + (cimg[1::2, 0::2] + cimg[1::2, 1::2])) * np.float32(0.25)
    got = small.cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    assert np.nanmax(np.abs(got - ref)) <= 5e-6
    # the filter itself is exact: the GPU's own large frame filtered in numpy gives the GPU's small frame bit for bit
    gbig = big.cpu().numpy()
    mine = ((gbig[0::2, 0::2] + gbig[0::2, 1::2]) + (gbig[1::2, 0::2] + gbig[1::2, 1::2])) * np.float32(0.25)
    assert got.tobytes() == mine.tobytes()
