"""GPU parity: the HIP path (through the C ABI of libvoxelhip.so) against the CPU oracle and the reference's goldens.

Bar: hit identity, step counts and every traversal float (t, pos, uv, per-iteration t_min) are
BIT-EXACT against the oracle; shaded colour within COLOR_TOL (libm vs ocml in pow/acos). The reference's own golden
vectors are replayed through the GPU with the tolerances the reference's tests state (1e-5).
"""
import os
from pathlib import Path

import numpy as np
import pytest

from helpers import ROOT, SVO_TYPES, build_world, golden_materials, golden_textures, oracle_scene, orc, vra

pytestmark = pytest.mark.gpu

FMTS = ["esvo", "csvo"]
EPS = 1e-5
COLOR_TOL = 5e-6  # absolute, on colours in [0,1]: powf/acosf differ by a few ulp between glibc and ocml, texel blending order


def f32(x):
    return np.asarray(x, dtype=np.float32).astype(np.float64)


@pytest.fixture(scope="module")
def hip():
    from voxel_rs_amd import hip as h

    return h


def gpu_svo(hip, golden, fmt, world, capacity=8 << 20):
    svo = hip.Svo(SVO_TYPES[fmt], capacity)
    svo.set_materials(golden_materials(golden))
    tex, mips = golden_textures(golden)
    svo.set_textures(tex, mips)
    svo.update(world)
    return svo


def same_bits(a, b):
    return np.array_equal(np.asarray(a, dtype=np.float32).view(np.uint32), np.asarray(b, dtype=np.float32).view(np.uint32))


def assert_result_equal(gpu, cpu, what):
    assert same_bits(gpu.t, cpu.t), (what, "t", gpu.t, cpu.t)
    assert gpu.value == cpu.value and gpu.face_id == cpu.face_id and bool(gpu.inside_voxel) == bool(cpu.inside_voxel), what
    assert same_bits(list(gpu.pos), list(cpu.pos)), (what, "pos", list(gpu.pos), list(cpu.pos))
    assert same_bits(list(gpu.uv), list(cpu.uv)), (what, "uv")
    assert same_bits(list(gpu.color), list(cpu.color)), (what, "color")


# ---- the reference's single-ray tests (svo_shader_tests.rs) replayed on the GPU -----------------------------------


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("case", ["shader_svo_traversal", "check_at_higher_coordinates"])
def test_traversal_frames(hip, golden, fmt, case):
    g = golden["formats"][fmt][case]
    scene, world = oracle_scene(golden, fmt, g["svo_pos"], g["blocks"])
    svo = gpu_svo(hip, golden, fmt, world)
    ray = g["ray"]
    d = orc.normalize(ray["dir"])
    res, frames, n = svo.debug_trace(ray["pos"], d, ray["max_dst"], ray["cast_translucent"], max_frames=100)
    cres, cframes, cn = scene.intersect(ray["pos"], d, ray["max_dst"], ray["cast_translucent"], max_frames=100)
    # against the reference's golden frames
    assert n == len(g["frames"])
    for i, (got, exp) in enumerate(zip(frames, g["frames"])):
        assert abs(float(got["t_min"]) - float(f32(exp["t_min"]))) < EPS, (i, got, exp)
        for k in ("ptr", "idx", "parent_octant_idx", "scale", "is_child", "is_leaf", "crossed_boundary"):
            assert int(got[k]) == exp[k], (i, k, got, exp)
        if fmt == "csvo":
            assert int(got["next_ptr"]) == exp["next_ptr"], (i, got, exp)
    # and bit for bit against the oracle
    assert n == cn
    assert frames.tobytes() == cframes.tobytes()
    assert_result_equal(res, cres, case)
    e = g["result"]
    assert abs(res.t - float(f32(e["t"]))) <= max(e["t_tol"], 0) + 1e-12
    np.testing.assert_allclose(list(res.pos), f32(e["pos"]), rtol=0, atol=max(e["pos_tol"], 1e-12))
    np.testing.assert_allclose(list(res.uv), f32(e["uv"]), rtol=0, atol=max(e["uv_tol"], 1e-12))
    np.testing.assert_allclose(list(res.color), f32(e["color"]), rtol=0, atol=1e-12)


@pytest.mark.parametrize("fmt", FMTS)
def test_golden_result_tables(hip, golden, fmt):
    """cast_inside_outside_all_axes, uv_coords_on_all_sides, casting_against_translucent_leafs, detect_inside_leaf_voxel."""
    G = golden["formats"][fmt]
    for name in ("cast_inside_outside_all_axes", "uv_coords_on_all_sides", "casting_against_translucent_leafs", "detect_inside_leaf_voxel"):
        g = G[name]
        scene, world = oracle_scene(golden, fmt, g["svo_pos"], g["blocks"])
        svo = gpu_svo(hip, golden, fmt, world)
        for case in g["cases"]:
            d = orc.normalize(case["dir"])
            ct = case.get("cast_translucent", g.get("cast_translucent", False))
            res, _, _ = svo.debug_trace(case["pos"], d, g["max_dst"], ct, max_frames=0)
            cres, _, _ = scene.intersect(case["pos"], d, g["max_dst"], ct)
            assert_result_equal(res, cres, (name, case.get("name")))
            if "expected" in case:
                e = case["expected"]
                tol = lambda k: max(e[k], EPS if name == "cast_inside_outside_all_axes" else 0.0) + 1e-12  # noqa: E731
                assert abs(res.t - float(f32(e["t"]))) <= tol("t_tol"), (name, case.get("name"))
                assert res.value == e["value"] and res.face_id == e["face_id"] and bool(res.inside_voxel) == e["inside_voxel"]
                np.testing.assert_allclose(list(res.pos), f32(e["pos"]), rtol=0, atol=tol("pos_tol"))
                np.testing.assert_allclose(list(res.uv), f32(e["uv"]), rtol=0, atol=tol("uv_tol"))
                np.testing.assert_allclose(list(res.color), f32(e["color"]), rtol=0, atol=tol("color_tol"))
            else:
                np.testing.assert_allclose(list(res.uv), f32(case["expected_uv"]), rtol=0, atol=EPS)
                np.testing.assert_allclose(list(res.color), f32(case["expected_color"]), rtol=0, atol=EPS)
            if name == "cast_inside_outside_all_axes":
                # printed golden floats are reproduced bit for bit
                assert same_bits(res.t, e["t"]) and same_bits(list(res.pos), e["pos"]) and same_bits(list(res.uv), e["uv"]), case["name"]


@pytest.mark.parametrize("fmt", FMTS)
def test_picker_end_to_end(hip, golden, fmt):
    """src/graphics/svo.rs:402-449 through vx_raycast."""
    g = golden["picker_raycast"]
    scene, world = oracle_scene(golden, fmt, [0, 0, 0], g["blocks"], compact=g["compact_chunk"])
    svo = gpu_svo(hip, golden, fmt, world)
    tasks = np.zeros(len(g["rays"]), dtype=hip.PICKER_TASK_DTYPE)
    for i, r in enumerate(g["rays"]):
        tasks[i]["max_dst"], tasks[i]["pos"], tasks[i]["dir"] = r["max_dst"], r["pos"], r["dir"]
    out = svo.raycast(tasks)
    for got, exp in zip(out, g["expected"]):
        assert abs(float(got["dst"]) - exp["dst"]) < g["tol"]
        assert bool(got["inside_voxel"]) == exp["inside_voxel"]
        np.testing.assert_allclose(got["pos"], exp["pos"], rtol=0, atol=g["tol"])
        np.testing.assert_array_equal(got["normal"], np.asarray(exp["normal"], dtype=np.float32))
    assert out.tobytes() == scene.picker(tasks.view(orc.PICKER_TASK_DTYPE)).tobytes()


# ---- seeded random worlds: picker rays, GPU vs oracle, bit-exact ---------------------------------------------------


def random_chunk_blocks(rng, n, ids):
    pts = rng.integers(0, 32, size=(n, 3))
    return [[int(x), int(y), int(z), int(rng.choice(ids))] for x, y, z in pts]


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("seed,svo_pos,n_blocks", [(1, (0, 0, 0), 40), (2, (1, 0, 1), 600), (3, (3, 2, 1), 6000), (4, (0, 0, 0), 0)])
def test_random_world_picker_rays(hip, golden, fmt, seed, svo_pos, n_blocks):
    rng = np.random.default_rng(seed)
    blocks = random_chunk_blocks(rng, n_blocks, [1, 2, 3, 4])
    if n_blocks == 0:
        blocks = [[5, 5, 5, 1]]  # an (almost) empty world: nearly every ray misses
    scene, world = oracle_scene(golden, fmt, svo_pos, blocks)
    svo = gpu_svo(hip, golden, fmt, world)
    n = 4096
    size = float(1 << world.depth)
    tasks = np.zeros(n, dtype=hip.PICKER_TASK_DTYPE)
    origin = rng.uniform(-8.0, size + 8.0, size=(n, 3)).astype(np.float32)
    target = (np.asarray(svo_pos, dtype=np.float32) * 32 + rng.uniform(0, 32, size=(n, 3))).astype(np.float32)
    d = target - origin
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::17] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, size=len(d[::17]))] * rng.choice([-1.0, 1.0], size=(len(d[::17]), 1))  # axis-aligned rays (zero components)
    tasks["pos"], tasks["dir"] = origin, d.astype(np.float32)
    tasks["max_dst"] = np.where(rng.random(n) < 0.3, rng.uniform(1, 40, size=n), -1.0).astype(np.float32)
    got = svo.raycast(tasks)
    exp = scene.picker(tasks.view(orc.PICKER_TASK_DTYPE), threads=4)
    assert got.tobytes() == exp.tobytes()
    assert (exp["dst"] > 0).sum() > (0 if n_blocks == 0 else 50)  # the comparison above is not vacuous
    # origins inside the chunk: with 6000 blocks about one ray in seven starts INSIDE a voxel, where the reference keeps
    # descending below the leaf level (svo.esvo.glsl:183-185) -- the stack's spill levels are what this exercises
    tasks["pos"] = (np.asarray(svo_pos, dtype=np.float32) * 32 + rng.uniform(0, 32, size=(n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3))
    tasks["dir"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    tasks["max_dst"] = -1.0
    got = svo.raycast(tasks)
    exp = scene.picker(tasks.view(orc.PICKER_TASK_DTYPE), threads=4)
    assert got.tobytes() == exp.tobytes()
    if n_blocks >= 6000:
        assert exp["inside_voxel"].sum() > 100


# ---- full frames: shading, shadows, translucency, trilinear sampling -------------------------------------------------


def compare_frames(img, hits, cimg, chits):
    for k in ("t", "pos", "uv", "lod", "shadow_t"):
        assert same_bits(hits[k], chits[k]), k
    for k in ("value", "face_id", "flags", "steps"):
        assert np.array_equal(hits[k], chits[k]), k
    nan_g, nan_c = np.isnan(img), np.isnan(cimg)
    assert np.array_equal(nan_g, nan_c)  # sky colour is NaN where acos' argument exceeds 1 (world.glsl:98), on both sides
    diff = np.abs(np.where(nan_g, 0, img) - np.where(nan_c, 0, cimg))
    assert diff.max() <= COLOR_TOL, diff.max()


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("depth,size", [(7, (160, 96)), (9, (320, 180))])
def test_heightfield_frame(hip, fmt, depth, size):
    from voxel_rs_amd import scenes

    world = vra.World(SVO_TYPES[fmt])
    st = world.build_heightfield(depth, threads=4)
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    scene = orc.OracleScene(SVO_TYPES[fmt], world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    svo = hip.Svo(SVO_TYPES[fmt], world.size_in_bytes + (1 << 20))
    svo.set_materials(mats)
    svo.set_textures(tex, 6)
    svo.update(world)
    assert svo.get_stats()["depth"] == depth
    w, h = size
    for shadows, sd in ((True, 500.0), (False, 500.0), (True, 40.0)):
        u = scenes.bench_camera(depth, st["h_max"], w, h, shadow_distance=sd, render_shadows=shadows)
        img, hits = svo.render(u, w, h, want_hits=True)
        cimg, chits = scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), w, h)
        compare_frames(img, hits, cimg, chits)
        assert (hits["flags"] & 1).mean() > 0.2
        img2, _ = svo.render(u, w, h, want_hits=False)  # the kernel variant without hit records
        assert img2.tobytes() == img.tobytes()
        # instrumented variant counts exactly what the oracle counts
        oc = orc.Counters()
        scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), w, h, want_hits=False, counters=oc)
        gc = svo.render_counters(u, w, h)
        for k, v in oc.as_dict().items():
            assert gc[k] == v, (k, gc[k], v)
        assert gc["pixels"] == w * h


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("tex_size,levels", [(48, 5), (40, 4), (24, 4)])
def test_textures_whose_height_is_not_a_power_of_two(hip, fmt, tex_size, levels):
    """WRAP_T is REPEAT (texture_array.rs never sets it): for heights that are not powers of two the sampler's wrap is a modulo, not a mask
    (TexLevel::repeat_t) -- frames, hit records and fetch counters against the oracle, with texture minification in play (lod > 0 from 15 blocks on)."""
    from voxel_rs_amd import scenes

    world = vra.World(SVO_TYPES[fmt])
    st = world.build_heightfield(8, threads=4)
    tex, mats = scenes.synthetic_textures(size=tex_size), scenes.synthetic_materials()
    scene = orc.OracleScene(SVO_TYPES[fmt], world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, levels)
    svo = hip.Svo(SVO_TYPES[fmt], world.size_in_bytes + (1 << 20))
    svo.set_materials(mats)
    svo.set_textures(tex, levels)
    svo.update(world)
    w, h = 224, 128
    u = scenes.bench_camera(8, st["h_max"], w, h, shadow_distance=500.0, render_shadows=True)
    img, hits = svo.render(u, w, h, want_hits=True)
    cimg, chits = scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), w, h)
    compare_frames(img, hits, cimg, chits)
    assert (hits["lod"] > 0).mean() > 0.2  # (minified samples: both mip levels' wraps are exercised)
    img2, _ = svo.render(u, w, h, want_hits=False)
    assert img2.tobytes() == img.tobytes()


@pytest.mark.parametrize("base,depth", [((200, 3, 201), 13), ((400, 3, 401), 14), ((800, 3, 801), 15)])
@pytest.mark.parametrize("fmt", FMTS)
def test_rays_from_inside_voxels_in_a_deep_world(hip, fmt, base, depth, monkeypatch):
    """Depth 13 (a few chunks far from the origin): the leaves sit on the last LDS-resident stack level, so a ray that
    starts inside a voxel (every primary ray of a camera buried in a block, and shadow rays that start inside a
    neighbour) is led below them by leaf data (svo.esvo.glsl:183-185 only accepts a leaf when t_min > 0) and has to be
    carried through the full, spill-backed stack. Depth 14: ordinary descents of the world's own bytes leave the LDS-resident
    levels too; depth 15: so do descents of a CSVO world's traversal image (the kernel variant with the hand-over test)."""
    import math
    from voxel_rs_amd import scenes

    rng = np.random.default_rng(11)
    world = vra.World(SVO_TYPES[fmt])
    for dx in range(2):
        for dz in range(2):
            chunk = vra.Chunk(base[0] + dx, base[1], base[2] + dz, 5)
            for x in range(32):
                for z in range(32):
                    top = 6 + int(rng.integers(0, 6))
                    for y in range(top):
                        chunk.set_block(x, y, z, int(rng.choice([1, 2, 3, 7, 9])))
            for _ in range(200):
                x, y, z = (int(v) for v in rng.integers(0, 32, size=3))
                chunk.set_block(x, y, z, int(rng.choice([5, 10, 4])))
            chunk.compact()
            world.set_chunk((base[0] + dx, base[1], base[2] + dz), chunk)
    world.serialize()
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    scene = orc.OracleScene(SVO_TYPES[fmt], world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    # a frame whose eye is inside a voxel normally goes straight to the kernel that traverses the world's own bytes; at depth 13
    # it is kept on the image kernel, whose inside-voxel hand-over then has every primary ray to deal with
    monkeypatch.setenv("VX_EYE_CHECK", "0" if depth == 13 else "1")
    svo = hip.Svo(SVO_TYPES[fmt], world.size_in_bytes + (1 << 20))
    svo.set_materials(mats)
    svo.set_textures(tex, 6)
    svo.update(world)
    assert svo.get_stats()["depth"] == depth
    w, h = 96, 64
    ox, oy, oz = (32.0 * c for c in base)
    wandering = 0
    for eye, fwd in (((ox + 20.3, oy + 2.4, oz + 30.6), (0.6, 0.35, 0.7)), ((ox + 33.5, oy + 4.5, oz + 12.5), (0.2, 0.9, -0.3)),
                     ((ox + 9.5, oy + 1.5, oz + 40.5), (-0.5, 0.1, 0.8)), ((ox + 30.0, oy + 20.0, oz + 30.0), (0.5, -0.6, 0.6))):
        u = scenes.render_params_to_uniforms(eye, fwd, (0.0, 1.0, 0.0), math.radians(72.0), w / h, 0.3, (-1.0, -1.0, -1.0), True, 500.0)
        img, hits = svo.render(u, w, h, want_hits=True)
        cimg, chits = scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), w, h)
        compare_frames(img, hits, cimg, chits)
        oc = orc.Counters()
        scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), w, h, want_hits=False, counters=oc)
        gc = svo.render_counters(u, w, h)
        for k, v in oc.as_dict().items():
            assert gc[k] == v, (k, gc[k], v)
        wandering += int((chits["steps"] > 3 * 13).sum())
    assert wandering > 100  # not vacuous: many rays go well past a plain root-to-leaf descent
    # Frame after frame of the last view, a few in flight: with this many inside-voxel rays the renderer moves between the image
    # kernel and the kernel on the world's own bytes as it goes (vx_api.hip, "steering"); every frame is the same frame.
    import torch

    targets = [torch.zeros((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    for i in range(70):
        svo.render_device(u, w, h, targets[i % 3].data_ptr())
        if i % 3 == 2:
            svo.sync()
            for t in targets:
                got = t.cpu().numpy()
                assert np.array_equal(np.isnan(got), np.isnan(img)) and np.nanmax(np.abs(got - img)) == 0.0


@pytest.mark.parametrize("fmt", FMTS)
def test_translucent_blocks_frame(hip, fmt):
    """Glass panes and leaves in front of terrain: the adjacency/translucency rule (svo.esvo.glsl:241-265) in a frame,
    a highlighted voxel (world.glsl:37-45) and a camera inside the domain looking at nearby geometry (NEAREST sampling)."""
    from voxel_rs_amd import scenes

    chunk = vra.Chunk(0, 0, 0, 5)
    for x in range(32):
        for z in range(32):
            chunk.set_block(x, 0, z, 3 if (x + z) % 3 else 7)
    for x in range(4, 28):
        for y in range(1, 9):
            chunk.set_block(x, y, 10, 5)   # glass wall
            chunk.set_block(x, y, 11, 5)   # second identical layer: skipped as "not first of its kind"
            chunk.set_block(x, y, 14, 10)  # leaves
    for y in range(1, 12):
        chunk.set_block(16, y, 20, 9)
    chunk.compact()
    world = vra.World(SVO_TYPES[fmt])
    world.set_chunk((0, 0, 0), chunk)
    world.serialize()
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    scene = orc.OracleScene(SVO_TYPES[fmt], world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    svo = hip.Svo(SVO_TYPES[fmt], 4 << 20)
    svo.set_materials(mats)
    svo.set_textures(tex, 6)
    svo.update(world)
    w, h = 256, 160
    u = scenes.render_params_to_uniforms((16.0, 6.0, -6.0), (0.0, -0.1, 1.0), (0.0, 1.0, 0.0), np.radians(72.0), w / h, selected_voxel=(10.0, 3.0, 10.0))
    img, hits = svo.render(u, w, h, want_hits=True)
    cimg, chits = scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), w, h)
    compare_frames(img, hits, cimg, chits)
    assert (hits["flags"] & 8).any(), "highlight outline not exercised"
    assert (hits["value"] == 9).any() and (hits["value"] == 5).any()


# ---- incremental update, tile sharding, error behaviour ------------------------------------------------------------------


@pytest.mark.parametrize("fmt", FMTS)
def test_incremental_update_ranges(hip, golden, fmt):
    """Svo::update with dirty ranges only (svo.rs:171-189): add a second chunk, re-commit, compare with a fresh oracle."""
    c0 = vra.Chunk(0, 0, 0, 5)
    c0.apply_blocks([dict(box=[[0, 32], [0, 2], [0, 32]], id=1)])
    c0.compact()
    world = vra.World(SVO_TYPES[fmt])
    world.set_chunk((0, 0, 0), c0)
    world.serialize()
    svo = gpu_svo(hip, golden, fmt, world)
    c1 = vra.Chunk(1, 0, 0, 5)
    c1.apply_blocks([dict(box=[[0, 32], [0, 6], [0, 32]], id=2)])
    c1.compact()
    world.set_chunk((1, 0, 0), c1)
    world.serialize()
    ranges = world.updated_ranges()
    assert ranges and sum(n for _, n in ranges) < world.size_in_bytes + 1
    svo.update(world)
    tex, mips = golden_textures(golden)
    scene = orc.OracleScene(SVO_TYPES[fmt], world.frame(), golden_materials(golden), tex, mips)
    rng = np.random.default_rng(7)
    n = 2048
    tasks = np.zeros(n, dtype=hip.PICKER_TASK_DTYPE)
    tasks["pos"] = rng.uniform(0, 64, size=(n, 3)).astype(np.float32) * np.float32([1, 0.3, 0.5]) + np.float32([0, 8, 0])
    tasks["dir"] = np.float32([0, -1, 0])
    tasks["max_dst"] = -1.0
    got = svo.raycast(tasks)
    assert got.tobytes() == scene.picker(tasks.view(orc.PICKER_TASK_DTYPE)).tobytes()
    top = got["pos"][:, 1][got["dst"] > 0]
    assert set(np.round(top).astype(int)) == {2, 6}


@pytest.mark.parametrize("fmt", FMTS)
def test_pipelined_commits_show_whole_versions(hip, fmt):
    """VX_COMMIT_PIPELINED: vx_commit posts the job and returns; frames rendered meanwhile show the world of the commit before or
    the new one -- never a mixture of the two (world bytes, image and origin table change together) --, and after vx_commit_wait
    the new one. Every frame is compared with the oracle's frames of both versions."""
    from voxel_rs_amd import scenes

    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    w, h = 192, 128
    u = scenes.render_params_to_uniforms((30.0, 26.0, -14.0), (0.05, -0.45, 1.0), (0.0, 1.0, 0.0), np.radians(72.0), w / h)
    ou = orc.Uniforms.from_buffer_copy(bytes(u))
    world = vra.World(SVO_TYPES[fmt])
    floor = vra.Chunk(0, 0, 0, 5)
    floor.apply_blocks([dict(box=[[0, 32], [0, 2], [0, 32]], id=1)])
    floor.compact()
    world.set_chunk((0, 0, 0), floor)
    world.set_chunk((1, 0, 0), floor_at(1))
    world.serialize()
    svo = hip.Svo(SVO_TYPES[fmt], 8 << 20)
    svo.set_materials(mats)
    svo.set_textures(tex, 6)
    svo.update(world)  # the first commit of a context is done inline in either mode
    svo.set_commit_mode(True)

    def oracle_frame():
        scene = orc.OracleScene(SVO_TYPES[fmt], world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
        return scene.render(ou, w, h)

    old = oracle_frame()
    img, hits = svo.render(u, w, h, want_hits=True)
    compare_frames(img, hits, *old)
    saw_old = 0
    for step in range(6):
        # a tower of a different height and block every step: a third of the pixels change
        c = vra.Chunk(0, 0, 0, 5)
        c.apply_blocks([dict(box=[[0, 32], [0, 2], [0, 32]], id=1), dict(box=[[4, 28], [2, 6 + 4 * step], [8, 24]], id=2 + step % 3)])
        c.compact()
        world.set_chunk((0, 0, 0), c)
        world.serialize()
        new = oracle_frame()
        assert not np.array_equal(new[1]["value"], old[1]["value"])
        svo.update(world)  # posts the job
        for _ in range(3):
            img, hits = svo.render(u, w, h, want_hits=True)
            if np.array_equal(hits["value"], old[1]["value"]):
                compare_frames(img, hits, *old)
                saw_old += 1
            else:
                compare_frames(img, hits, *new)
        svo.commit_wait()
        img, hits = svo.render(u, w, h, want_hits=True)
        compare_frames(img, hits, *new)
        old = new
    # back to inline commits: the worker is gone, an update is visible at once
    svo.set_commit_mode(False)
    world.set_chunk((0, 0, 0), floor)
    world.serialize()
    svo.update(world)
    img, hits = svo.render(u, w, h, want_hits=True)
    compare_frames(img, hits, *oracle_frame())
    print("frames that still showed the previous version:", saw_old)


def floor_at(cx):
    c = vra.Chunk(cx, 0, 0, 5)
    c.apply_blocks([dict(box=[[0, 32], [0, 3], [0, 32]], id=3)])
    c.compact()
    return c


@pytest.mark.parametrize("fmt", FMTS)
def test_pipelined_commits_under_frames_in_flight(hip, fmt):
    """Many pipelined commits while frames are queued on the frame streams without waiting (the streaming loop): no error, and
    the frame after the last commit is the oracle's."""
    import torch
    from voxel_rs_amd import scenes

    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    w, h = 256, 160
    u = scenes.render_params_to_uniforms((30.0, 30.0, -10.0), (0.05, -0.5, 1.0), (0.0, 1.0, 0.0), np.radians(72.0), w / h)
    world = vra.World(SVO_TYPES[fmt])
    for cx in range(2):
        world.set_chunk((cx, 0, 0), floor_at(cx))
    world.serialize()
    svo = hip.Svo(SVO_TYPES[fmt], 16 << 20)
    svo.set_materials(mats)
    svo.set_textures(tex, 6)
    svo.update(world)
    svo.set_commit_mode(True)
    svo.set_frames_in_flight(4)
    out = torch.zeros((4, h, w, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    rng = np.random.default_rng(3)
    for step in range(40):
        c = vra.Chunk(int(step % 2), 0, 0, 5)
        x0, z0 = (int(v) for v in rng.integers(0, 20, 2))
        c.apply_blocks([dict(box=[[0, 32], [0, 3], [0, 32]], id=3), dict(box=[[x0, x0 + 10], [3, 4 + step % 20], [z0, z0 + 10]], id=1 + step % 4)])
        c.compact()
        world.set_chunk((int(step % 2), 0, 0), c)
        world.serialize()
        svo.update(world)
        for k in range(2):
            svo.render_device(u, w, h, out[(2 * step + k) % 4].data_ptr())
    svo.commit_wait()
    img, hits = svo.render(u, w, h, want_hits=True)
    scene = orc.OracleScene(SVO_TYPES[fmt], world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    compare_frames(img, hits, *scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), w, h))
    svo.sync()


@pytest.mark.parametrize("fmt", FMTS)
def test_image_sizes_around_the_queue_edges(hip, fmt):
    """The sub-tile queue at its edges: images of one pixel, of less than a sub-tile, of exactly one tile, a pixel more than a tile, fewer
    sub-tiles than waves and more, rendered one after the other on the same context (its two sets of ticket dispensers take turns
    and clear each other), image-only and with hit records, one frame at a time and on the frame streams -- against the oracle."""
    import torch
    from voxel_rs_amd import scenes

    world = vra.World(SVO_TYPES[fmt])
    st = world.build_heightfield(7, threads=4)
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    scene = orc.OracleScene(SVO_TYPES[fmt], world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    svo = hip.Svo(SVO_TYPES[fmt], world.size_in_bytes + (1 << 20))
    svo.set_materials(mats)
    svo.set_textures(tex, 6)
    svo.update(world)
    for w, h in ((1, 1), (7, 3), (8, 8), (32, 32), (33, 33), (64, 8), (300, 260), (8, 200), (1, 1)):
        u = scenes.bench_camera(7, st["h_max"], w, h)
        cimg, chits = scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), w, h)
        img, hits = svo.render(u, w, h, want_hits=True)
        compare_frames(img, hits, cimg, chits)
        img2, _ = svo.render(u, w, h)  # image-only kernel, the context's own stream
        assert img2.tobytes() == img.tobytes()
        out = torch.zeros((3, h, w, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        for k in range(3):  # frame streams, frames in flight
            svo.render_device(u, w, h, out[k].data_ptr())
        svo.sync()
        for k in range(3):
            assert out[k].cpu().numpy().tobytes() == img.tobytes(), (w, h, k)


@pytest.mark.parametrize("fmt", FMTS)
def test_tile_sharded_render_matches_full(hip, fmt):
    import ctypes as C

    import torch
    from voxel_rs_amd import scenes

    world = vra.World(SVO_TYPES[fmt])
    st = world.build_heightfield(8, threads=4)
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    svo = hip.Svo(SVO_TYPES[fmt], world.size_in_bytes + (1 << 20))
    svo.set_materials(mats)
    svo.set_textures(tex, 6)
    svo.update(world)
    w, h = 200, 120  # not a multiple of the 32-pixel tile
    u = scenes.bench_camera(8, st["h_max"], w, h)
    full, _ = svo.render(u, w, h)
    count = 3
    per = max(hip.local_tile_count(w, h, r, count) for r in range(count))
    gathered = torch.zeros((count, per, 32, 32, 4), dtype=torch.float32, device="cuda")
    out = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()  # torch zero-fills on ITS stream; the renderer's streams do not wait for it
    for r in range(count):
        svo.render_device(u, w, h, gathered[r].data_ptr(), tile_rank=r, tile_count=count)
    svo.assemble_tiles(gathered.data_ptr(), per * 32 * 32 * 4, count, w, h, out.data_ptr())
    svo.sync()
    got = out.cpu().numpy()
    assert got.tobytes() == full.tobytes()
    # the host-memory flavour of a sharded render returns the same compact tile list
    tiles, _ = svo.render(u, w, h, tile_rank=1, tile_count=count)
    assert tiles.tobytes() == gathered[1, : tiles.shape[0]].cpu().numpy().tobytes()
    del C


def test_error_behaviour(hip):
    """No panics across the ABI: capacity overflow and misuse come back as error codes (svo.rs panics, esvo.rs:328 asserts)."""
    svo = hip.Svo(1, 1 << 16)
    with pytest.raises(hip.VoxelHipError, match="no SVO committed"):
        svo.render(hip.make_uniforms(np.eye(4, dtype=np.float32).ravel(), 1.0, 1.0, 0.3, (0, -1, 0), (0, 0, 0), 0, 1.0), 8, 8)
    r = (hip.Range * 1)()
    r[0].start, r[0].length = 1 << 15, 1 << 16
    rc = hip.lib().vx_commit(svo._h, 6, r, 1, 0)
    assert rc == 4 and b"not large enough" in hip.lib().vx_last_error()
    with pytest.raises(hip.VoxelHipError):
        hip.Svo(3, 1 << 16)
    # the wave slots a context leaves to RCCL's kernels: 0 .. 8, read per launch (vx_set_comm_headroom), visible through vx_debug_knobs
    assert svo.knobs()["comm_headroom"] == 4 and svo.knobs()["measurement_build"] == 0
    svo.set_comm_headroom(0)
    assert svo.knobs()["comm_headroom"] == 0
    svo.set_comm_headroom(8)
    assert svo.knobs()["comm_headroom"] == 8
    with pytest.raises(hip.VoxelHipError, match="0..8"):
        svo.set_comm_headroom(9)
    svo.close()


OCCUPANCY_COUNTERS = ("wave_steps", "services", "refills", "tail_wave_steps", "tail_iterations")


@pytest.mark.parametrize("fmt", FMTS)
def test_kernel_versions_agree(hip, fmt, monkeypatch):
    """Every build of the render kernel a knob of the product library selects, each against the ORACLE (hit records and step counts bit for
    bit, colours within COLOR_TOL, the instrumented counters equal to the oracle's): the persistent wavefront kernel (default), the
    one-thread-per-pixel kernel, service thresholds (they only reorder work between lanes), the traversal image (default) and the world's own
    bytes, the image layout for more than 4 GiB (octant indices behind a 64-bit pointer), the LDS copy of the top levels (VX_HOT_LEVELS:
    north_star's "hot upper octree levels staged in LDS" -- a depth-10 world, so that there are levels below the staged ones), the walk
    inside voxels in the render loop of a small world, three frames in flight. A leg's image-only frames (the builds without hit records:
    the hot-levels build is one) must be the frame that was checked, byte for byte. (The library's measurement build and its knobs:
    test_timeline_build_renders_the_same_frames, test_knobs_of_the_measurement_build_change_no_pixel.)"""
    from voxel_rs_amd import scenes

    depth = 10
    world = vra.World(SVO_TYPES[fmt])
    st = world.build_heightfield(depth, threads=4)
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    w, h = 250, 130
    u = scenes.bench_camera(depth, st["h_max"], w, h)
    # the checker: one render of the view, and what it counted
    scene = orc.OracleScene(SVO_TYPES[fmt], world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    ou = orc.Uniforms.from_buffer_copy(bytes(u))
    cimg, chits = scene.render(ou, w, h)
    oc = orc.Counters()
    scene.render(ou, w, h, want_hits=False, counters=oc)
    assert (chits["flags"] & 1).mean() > 0.2 and ((chits["flags"] >> 1) & 1).sum() > 1000
    results = []
    knobs = ("VX_RENDER_KERNEL", "VX_SERVICE_MIN", "VX_TRAVERSAL_IMAGE", "VX_WIDE_IMAGE", "VX_HOT_LEVELS", "VX_FRAMES_IN_FLIGHT", "VX_FOREIGN_RERUN")
    # (the knobs of experiments -- the refill threshold, the order table off -- exist in the library's measurement build only:
    # test_knobs_of_the_measurement_build_change_no_pixel)
    for env in ({"VX_RENDER_KERNEL": "1"}, {}, {"VX_SERVICE_MIN": "1"}, {"VX_SERVICE_MIN": "33"},
                {"VX_TRAVERSAL_IMAGE": "0"}, {"VX_WIDE_IMAGE": "1"}, {"VX_HOT_LEVELS": "1"}, {"VX_FOREIGN_RERUN": "0"},
                {"VX_FRAMES_IN_FLIGHT": "3"}):
        for k in knobs:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        svo = hip.Svo(SVO_TYPES[fmt], world.size_in_bytes + (1 << 20))
        svo.set_materials(mats)
        svo.set_textures(tex, 6)
        svo.update_full(world)  # a fresh buffer each time: the world's dirty ranges were consumed by the first update
        img, hits = svo.render(u, w, h, want_hits=True)
        for frame in range(7):  # (from the third frame of a view on, the queue hands the sub-tiles out by last frame's cost, and the passes are last frames' sorted ones)
            img2, _ = svo.render(u, w, h)
            if img2.tobytes() != img.tobytes():
                bad = np.argwhere((img2.view(np.uint32) != img.view(np.uint32)).any(axis=2))
                y, x = bad[0]
                raise AssertionError(f"{env}: image-only frame {frame} differs from the frame with hit records in {len(bad)} pixels, first (x={x}, y={y}): "
                                     f"{img2[y, x]} against {img[y, x]}; {bad[:8].tolist()}")
        # (the occupancy counters describe how a kernel scheduled its lanes, not what the rays did)
        counters = {k: v for k, v in svo.render_counters(u, w, h).items() if k not in OCCUPANCY_COUNTERS}
        svo.close()
        # ... against the oracle: this leg's frame, hit records and counters
        try:
            compare_frames(img, hits, cimg, chits)
        except AssertionError as e:
            raise AssertionError(f"{env}: differs from the oracle: {e}") from e
        for k, v in oc.as_dict().items():
            assert counters[k] == v, (env, k, counters[k], v)
        results.append((img, hits.tobytes(), counters))
    for r in results[1:]:  # (follows from the above; kept: the builds among themselves agree more closely than with the oracle's libm)
        assert r[1] == results[0][1], "hit records differ between kernel versions"
        assert r[2] == results[0][2], "step counters differ between kernel versions"


@pytest.mark.parametrize("fmt", FMTS)
def test_timeline_build_renders_the_same_frames(hip, fmt):
    """The library's timeline build (voxel-rs_amd/lib/lib_tl: the image-only kernels stamp every wave's life, profiles/timeline.py) renders
    the frames of the plain library, byte for byte, and fills in its rows: run in a process of its own (a process loads one of the two)."""
    import hashlib
    import json
    import subprocess
    import sys

    from voxel_rs_amd import scenes

    world = vra.World(SVO_TYPES[fmt])
    st = world.build_heightfield(8, threads=4)
    svo = hip.Svo(SVO_TYPES[fmt], world.size_in_bytes + (1 << 20))
    svo.set_materials(scenes.synthetic_materials())
    svo.set_textures(scenes.synthetic_textures(), 6)
    svo.update_full(world)
    w, h = 250, 130
    u = scenes.bench_camera(8, st["h_max"], w, h)
    img, _ = svo.render(u, w, h)
    svo.close()
    code = f"""
import hashlib, json, os, sys
sys.path.insert(0, {str(ROOT)!r})
from _pkg import load_package
vra = load_package()
from voxel_rs_amd import hip, scenes
import torch
world = vra.World({SVO_TYPES[fmt]}); st = world.build_heightfield(8, threads=4)
svo = hip.Svo({SVO_TYPES[fmt]}, world.size_in_bytes + (1 << 20))
svo.set_materials(scenes.synthetic_materials()); svo.set_textures(scenes.synthetic_textures(), 6); svo.update_full(world)
svo.set_frames_in_flight(1)
u = scenes.bench_camera(8, st["h_max"], {w}, {h})
image = torch.zeros(({h}, {w}, 4), dtype=torch.float32, device="cuda")
for _ in range(3): svo.render_device(u, {w}, {h}, image.data_ptr())
svo.sync()
t = svo.timeline()
print(json.dumps(dict(sha=hashlib.sha256(image.cpu().numpy().tobytes()).hexdigest(), waves=int(len(t)), lived=int((t[:, 2] > t[:, 0]).sum()), trips=int(t[:, 6].sum()))))
"""
    env = dict(os.environ, VX_TIMELINE="1", VX_LIB_DIR=str(Path(ROOT) / "voxel-rs_amd" / "lib" / "lib_tl"))
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["sha"] == hashlib.sha256(img.tobytes()).hexdigest()
    assert d["waves"] > 0 and d["lived"] == d["waves"] and d["trips"] > d["waves"]


@pytest.mark.parametrize("fmt", FMTS)
def test_frames_of_a_moving_camera(hip, fmt):
    """The reference's frame loop moves the camera every frame (src/gamelogic/game.rs:111-148): thirty frames of a view that turns by a quarter of
    a degree a frame and walks, rendered as the benchmark renders them -- image-only, device resident, two frames in flight -- are, every one,
    byte for byte the frame of the render with hit records (another build of the kernel, one frame at a time), and three of them within the
    colour tolerance of the oracle's with its hit records exactly."""
    import math

    import torch
    from voxel_rs_amd import scenes

    depth, w, h = 9, 480, 270
    world = vra.World(SVO_TYPES[fmt])
    st = world.build_heightfield(depth, threads=4)
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    svo = hip.Svo(SVO_TYPES[fmt], world.size_in_bytes + (1 << 20))
    svo.set_materials(mats)
    svo.set_textures(tex, 6)
    svo.update_full(world)
    scene = orc.OracleScene(SVO_TYPES[fmt], world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    n = float(1 << depth)

    def uniforms(i):
        a = math.radians(0.25 * i)
        fwd = (0.6 * math.cos(a) - 0.7 * math.sin(a), -0.35, 0.6 * math.sin(a) + 0.7 * math.cos(a))
        eye = (0.5 * n + 0.08 * i, st["h_max"] + 0.05 * n, 0.5 * n - 0.03 * i)
        return scenes.render_params_to_uniforms(eye, fwd, (0.0, 1.0, 0.0), math.radians(72.0), w / h, 0.3, (-1.0, -1.0, -1.0), True, 3.0e38)

    frames = 30
    us = [uniforms(i) for i in range(frames)]
    images = [torch.zeros((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(frames)]
    torch.cuda.synchronize()
    svo.set_frames_in_flight(2)
    for i in range(frames):
        svo.render_device(us[i], w, h, images[i].data_ptr())
    svo.sync()
    for i in range(frames):
        img, hits = svo.render(us[i], w, h, want_hits=True)
        got = images[i].cpu().numpy()
        assert got.tobytes() == img.tobytes(), f"frame {i} of the moving view differs from the render with hit records"
        if i in (0, 11, frames - 1):
            cimg, chits = scene.render(orc.Uniforms.from_buffer_copy(bytes(us[i])), w, h)
            assert hits.tobytes() == chits.tobytes(), f"frame {i}: hit records differ from the oracle's"
            assert np.nanmax(np.abs(got - cimg)) <= COLOR_TOL
    svo.close()


@pytest.mark.parametrize("fmt", FMTS)
def test_views_and_sizes_in_turn(hip, fmt):
    """Views that stand still, change, come back, in two sizes: whatever a stream keeps between its frames (the order table of a view's
    sub-tiles), every frame is the frame -- device-resident targets on the frame streams (two in flight) and host targets on the context's
    own stream, two views and two sizes in turn."""
    import torch
    from voxel_rs_amd import scenes

    world = vra.World(SVO_TYPES[fmt])
    st = world.build_heightfield(8, threads=4)
    svo = hip.Svo(SVO_TYPES[fmt], world.size_in_bytes + (1 << 20))
    svo.set_materials(scenes.synthetic_materials())
    svo.set_textures(scenes.synthetic_textures(), 6)
    svo.update(world)
    sizes = ((250, 130), (192, 160))
    views = {}
    for w, h in sizes:
        ua = scenes.bench_camera(8, st["h_max"], w, h)
        ub = scenes.render_params_to_uniforms((100.0, st["h_max"] + 20.0, 90.0), (-0.5, -0.4, 0.6), (0.0, 1.0, 0.0), 1.2, w / h, 0.3, (-1.0, -1.0, -1.0), True, 500.0)
        for name, u in (("a", ua), ("b", ub)):
            views[(name, w, h)] = (u, svo.render(u, w, h, want_hits=True)[0])
    svo.set_frames_in_flight(2)
    targets = {(w, h): [torch.zeros((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(2)] for w, h in sizes}
    sequence = [("a", 0)] * 7 + [("b", 0)] * 2 + [("a", 0)] * 6 + [("a", 1)] * 5 + [("b", 1)] * 6 + [("a", 0)] * 5
    for i, (name, size) in enumerate(sequence):
        w, h = sizes[size]
        u, ref = views[(name, w, h)]
        t = targets[(w, h)][i % 2]
        svo.render_device(u, w, h, t.data_ptr())
        if i % 3 == 2 or i == len(sequence) - 1:  # (frames stay in flight most of the time)
            svo.sync()
            assert t.cpu().numpy().tobytes() == ref.tobytes(), (i, name, size)
        img, _ = svo.render(u, w, h)  # the context's own stream has tables of its own
        assert img.tobytes() == ref.tobytes(), (i, name, size, "host target")
    svo.sync()
    svo.close()


def run_knob_worker(fmt, env, measurement_build, occupancy=False):
    """tests/knob_worker.py in a process of its own (a process loads one build of the library; the knobs are read when a context is created).
    occupancy=True: also the scheduling knobs the worker's context says it runs with (vx_debug_knobs)."""
    import json
    import subprocess
    import sys

    e = {k: v for k, v in os.environ.items() if not k.startswith("VX_")}
    e.update(env)
    if measurement_build:
        e["VX_LIB_DIR"] = str(Path(ROOT) / "voxel-rs_amd" / "lib" / "lib_tl")
    r = subprocess.run([sys.executable, str(Path(ROOT) / "tests" / "knob_worker.py"), fmt], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    out.pop("occupancy")
    knobs = out.pop("knobs")
    return (out, knobs) if occupancy else out


@pytest.mark.parametrize("fmt", FMTS)
def test_the_order_of_the_tiles_changes_no_pixel(hip, fmt):
    """Where on the screen the queue's tile numbers lie (VX_TILE_NUMBERING: rows, strips of eight columns, the stride) is the ORDER a frame's work is done in
    and nothing else: a frame of an odd size -- a last strip narrower than the others, edge tiles -- with hit records, image-only one frame at a time (cost
    order: seven frames) and on the frame streams, and a rank's share of a tile list, under every numbering, byte for byte the frame of the default. In the
    library's measurement build also the width of the strips (VX_TILE_STRIP) and how many sub-tiles in a row go to one dispenser (VX_QUEUE_STRIPE)."""
    ref = run_knob_worker(fmt, {}, False)
    assert ref["own 0"] == ref["own 6"]
    for numbering in (0, 2):
        assert run_knob_worker(fmt, {"VX_TILE_NUMBERING": str(numbering)}, False) == ref, numbering
    for numbering, strip, stripe in ((1, 1, 16), (1, 3, 1), (1, 8, 100), (1, 100, 16), (2, 8, 1), (0, 8, 100000)):
        got = run_knob_worker(fmt, {"VX_TILE_NUMBERING": str(numbering), "VX_TILE_STRIP": str(strip), "VX_QUEUE_STRIPE": str(stripe)}, True)
        assert got == ref, (numbering, strip, stripe)


@pytest.mark.parametrize("fmt", FMTS)
def test_knobs_of_the_measurement_build_change_no_pixel(hip, fmt):
    """The knobs of experiments -- honoured by the library's measurement build only (lib/lib_tl; the product build ignores them) -- reorder work and nothing
    else: the refill and service thresholds, the order table off, a cap on the resident waves. Frames, hit records and step counters of the product build."""
    ref = run_knob_worker(fmt, {}, False)
    for env in ({"VX_REFILL_MIN": "1", "VX_SERVICE_MIN": "1"}, {"VX_REFILL_MIN": "7", "VX_SERVICE_MIN": "33"}, {"VX_HOT_FIRST": "0"}, {"VX_WAVES_PER_CU": "3"}):
        assert run_knob_worker(fmt, env, True) == ref, env
    # ... and the product build does not read them: what each build's context says it runs with (vx_debug_knobs)
    ref2, k_ref = run_knob_worker(fmt, {}, False, occupancy=True)
    got, k_product = run_knob_worker(fmt, {"VX_REFILL_MIN": "1", "VX_WAVES_PER_CU": "1", "VX_QUEUE_STRIPE": "3", "VX_TILE_STRIP": "3", "VX_HOT_FIRST": "0"}, False, occupancy=True)
    assert got == ref and ref2 == ref
    assert k_product == k_ref and k_ref["measurement_build"] == 0 and (k_ref["refill_min"], k_ref["waves_per_cu"], k_ref["queue_stripe"], k_ref["hot_first"]) == (64, 0, 0, 1), (k_ref, k_product)
    _, k_measure = run_knob_worker(fmt, {"VX_REFILL_MIN": "1", "VX_WAVES_PER_CU": "3", "VX_QUEUE_STRIPE": "5", "VX_HOT_FIRST": "0"}, True, occupancy=True)
    assert k_measure["measurement_build"] == 1 and (k_measure["refill_min"], k_measure["waves_per_cu"], k_measure["queue_stripe"], k_measure["hot_first"]) == (1, 3, 5, 0), k_measure


# ---- output formats, presentation ring, lifetime, fall-back, the library's own gather ---------------------------------------------


def as_image(img):
    """Framebuffer::as_image (src/graphics/framebuffer.rs:97-111): RGBA8 read-back (clamp, round to nearest of 255 steps), flipped."""
    v = np.where(np.isnan(img), np.float32(0), img).astype(np.float32)
    v = np.clip(v, np.float32(0), np.float32(1))
    return (v * np.float32(255) + np.float32(0.5)).astype(np.uint8)[::-1]


def heightfield_svo(hip, fmt, depth=8):
    from voxel_rs_amd import scenes

    world = vra.World(SVO_TYPES[fmt])
    st = world.build_heightfield(depth, threads=4)
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    svo = hip.Svo(SVO_TYPES[fmt], world.size_in_bytes + (1 << 20))
    svo.set_materials(mats)
    svo.set_textures(tex, 6)
    svo.update(world)
    return world, st, svo, tex, mats


@pytest.mark.parametrize("fmt", FMTS)
@pytest.mark.parametrize("size", [(640, 360), (1000, 562)])
def test_cost_ordered_queue_hands_out_every_subtile(hip, fmt, size):
    """One frame at a time on the context's own stream the library hands a view's sub-tiles out most expensive first, through the table
    order_kernel makes from the frame before last (a stable counting sort, 16 classes). Every frame is rendered into a buffer that was
    ZEROED first: a sub-tile the table lost or listed twice would leave (or fight over) pixels, and every frame must be the first."""
    import torch
    from voxel_rs_amd import scenes

    world = vra.World(SVO_TYPES[fmt])
    st = world.build_heightfield(9, threads=4)
    svo = hip.Svo(SVO_TYPES[fmt], world.size_in_bytes + (1 << 20))
    svo.set_materials(scenes.synthetic_materials())
    svo.set_textures(scenes.synthetic_textures(), 6)
    svo.update_full(world)
    svo.set_frames_in_flight(1)
    w, h = size
    u = scenes.bench_camera(9, st["h_max"], w, h, shadow_distance=3.0e38)
    img = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    frames = []
    for _ in range(7):
        img.zero_()
        torch.cuda.synchronize()
        svo.render_device(u, w, h, img.data_ptr())
        svo.sync()
        frames.append(img.cpu().numpy().copy())
    assert np.isfinite(frames[0]).all() and (frames[0][..., 3] == 1.0).all()  # every pixel written (alpha 1: sky or a lit hit)
    for f in frames[1:]:
        assert f.tobytes() == frames[0].tobytes()
    svo.close()


@pytest.mark.parametrize("fmt", FMTS)
def test_rgba8_target_is_as_image_of_the_float_frame(hip, fmt):
    import torch
    from voxel_rs_amd import scenes

    world, st, svo, tex, mats = heightfield_svo(hip, fmt)
    w, h = 200, 120
    u = scenes.bench_camera(8, st["h_max"], w, h)
    img, _ = svo.render(u, w, h)
    img8, _ = svo.render(u, w, h, fmt=hip.VX_FORMAT_RGBA8)
    assert img8.dtype == np.uint8 and img8.tobytes() == as_image(img).tobytes()
    # against the oracle's frame read back the same way: equal but for colours within 5e-6 of a rounding step
    scene = orc.OracleScene(SVO_TYPES[fmt], world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    cimg, _ = scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), w, h, want_hits=False)
    d = np.abs(img8.astype(np.int32) - as_image(cimg).astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 1e-3
    # tile lists in RGBA8, assembled: the same image
    count = 3
    per = max(hip.local_tile_count(w, h, r, count) for r in range(count))
    gathered = torch.zeros((count, per, 32, 32), dtype=torch.int32, device="cuda")
    out = torch.zeros((h, w), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for r in range(count):
        svo.render_device(u, w, h, gathered[r].data_ptr(), tile_rank=r, tile_count=count, fmt=hip.VX_FORMAT_RGBA8)
    svo.stream_wait_render(svo.stream)
    svo.sync()
    svo.assemble_tiles_format(gathered.data_ptr(), per * 32 * 32, count, w, h, out.data_ptr(), hip.VX_FORMAT_RGBA8, svo.stream)
    svo.sync()
    assert out.cpu().numpy().view(np.uint8).reshape(h, w, 4).tobytes() == img8.tobytes()


@pytest.mark.parametrize("fmt", FMTS)
def test_presentation_ring(hip, fmt):
    """vx_present_begin / vx_present_wait: frame k+1 is begun before frame k is waited for; every frame is the frame."""
    from voxel_rs_amd import scenes

    world, st, svo, tex, mats = heightfield_svo(hip, fmt)
    w, h = 320, 180
    cams = [scenes.bench_camera(8, st["h_max"] + 3 * i, w, h) for i in range(9)]
    for out_fmt in (hip.VX_FORMAT_RGBA8, hip.VX_FORMAT_RGBA32F):
        expect = [svo.render(u, w, h, fmt=out_fmt)[0] for u in cams]
        slots = [svo.present_begin(cams[0], w, h, out_fmt)]
        for k in range(1, len(cams)):
            slots.append(svo.present_begin(cams[k], w, h, out_fmt))
            got = svo.present_wait(slots[k - 1], w, h, out_fmt)
            assert got.tobytes() == expect[k - 1].tobytes(), (out_fmt, k)
        assert svo.present_wait(slots[-1], w, h, out_fmt).tobytes() == expect[-1].tobytes()
        assert len(set(slots[:4])) == 4


@pytest.mark.parametrize("fmt", FMTS)
def test_swapping_materials_and_textures_with_eight_frames_in_flight(hip, fmt):
    """vx_set_materials / vx_set_textures while frames are in flight on all eight frame streams: the old tables stay alive until
    those frames are done (they are in the kernels' arguments), every frame shows exactly one generation of them."""
    import torch
    from voxel_rs_amd import scenes

    world, st, svo, tex, mats = heightfield_svo(hip, fmt)
    w, h = 480, 270
    u = scenes.bench_camera(8, st["h_max"], w, h)
    mats_b = mats.copy()
    mats_b["specular_strength"] = 0.9
    mats_b["tex_top"][1], mats_b["tex_side"][1] = mats["tex_side"][3], mats["tex_top"][3]  # grass wears stone
    tex_b = tex.copy()
    tex_b[..., 0] = tex[..., 2]  # red <- blue
    frame_a = svo.render(u, w, h)[0]
    svo.set_materials(mats_b)
    svo.set_textures(tex_b, 6)
    frame_b = svo.render(u, w, h)[0]
    assert np.nanmax(np.abs(frame_a - frame_b)) > 0.05
    svo.set_frames_in_flight(8)
    targets = [torch.zeros((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(16)]
    torch.cuda.synchronize()
    for rounds in range(3):
        svo.set_materials(mats)
        svo.set_textures(tex, 6)
        for t in targets[:8]:
            svo.render_device(u, w, h, t.data_ptr())
        svo.set_materials(mats_b)  # eight frames in flight on eight streams read the tables this replaces
        svo.set_textures(tex_b, 6)
        for t in targets[8:]:
            svo.render_device(u, w, h, t.data_ptr())
        svo.sync()
        for i, t in enumerate(targets):
            got, exp = t.cpu().numpy(), (frame_a if i < 8 else frame_b)
            assert np.array_equal(np.isnan(got), np.isnan(exp)) and np.nanmax(np.abs(got - exp)) == 0.0, (rounds, i)


@pytest.mark.parametrize("fmt", FMTS)
def test_image_that_does_not_fit_falls_back_to_the_worlds_bytes(hip, fmt, monkeypatch):
    """The traversal image is an accelerator: when its device allocation fails (here: capped), the context renders from the world's
    own bytes -- same frames -- and a later commit that fits brings the image back."""
    from voxel_rs_amd import scenes

    monkeypatch.setenv("VX_IMAGE_CAP_BYTES", "4096")
    world, st, svo, tex, mats = heightfield_svo(hip, fmt)
    assert svo.image_info()["layout"] == 0
    w, h = 160, 96
    u = scenes.bench_camera(8, st["h_max"], w, h)
    scene = orc.OracleScene(SVO_TYPES[fmt], world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    img, hits = svo.render(u, w, h, want_hits=True)
    cimg, chits = scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), w, h)
    compare_frames(img, hits, cimg, chits)
    # one more chunk, committed incrementally into the context that has no image: still the oracle's frame
    extra = vra.Chunk(1, 7, 1, 5)
    extra.apply_blocks([dict(box=[[0, 32], [0, 8], [0, 32]], id=3)])
    extra.compact()
    world.set_chunk((1, 7, 1), extra)
    world.serialize()
    svo.update(world)
    assert svo.image_info()["layout"] == 0
    scene = orc.OracleScene(SVO_TYPES[fmt], world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    img, hits = svo.render(u, w, h, want_hits=True)
    cimg, chits = scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), w, h)
    compare_frames(img, hits, cimg, chits)
    monkeypatch.delenv("VX_IMAGE_CAP_BYTES")
    svo2 = hip.Svo(SVO_TYPES[fmt], world.size_in_bytes + (1 << 20))
    svo2.set_materials(mats)
    svo2.set_textures(tex, 6)
    svo2.update_full(world)
    assert svo2.image_info()["layout"] == 1
    img2, hits2 = svo2.render(u, w, h, want_hits=True)
    assert hits2.tobytes() == hits.tobytes()


def test_the_librarys_gather_with_one_rank(hip):
    """vx_comm_init / vx_gather_tiles / vx_wait_gather on a one-rank communicator (RCCL opened at run time): the root's own share is
    its tile list; assembled on the communicator's stream it is the frame. (More ranks need more GPUs: the N > 1 arithmetic is
    covered on CPU by tests/test_sharding.py.)"""
    import torch
    from voxel_rs_amd import scenes

    world, st, svo, tex, mats = heightfield_svo(hip, "csvo")
    w, h = 200, 120
    u = scenes.bench_camera(8, st["h_max"], w, h)
    full, _ = svo.render(u, w, h)
    svo.comm_init(1, 0, hip.comm_unique_id())
    with pytest.raises(hip.VoxelHipError, match="already has a communicator"):
        svo.comm_init(1, 0, hip.comm_unique_id())
    # "two ranks" rendered by this one GPU, each share gathered separately through the communicator (root = the only rank)
    count = 2
    per = max(hip.local_tile_count(w, h, r, count) for r in range(count))
    lists = torch.zeros((count, per, 32, 32, 4), dtype=torch.float32, device="cuda")
    gathered = torch.zeros((count, per, 32, 32, 4), dtype=torch.float32, device="cuda")
    out = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    tickets = {}
    for frame in range(3):
        for r in range(count):
            if r in tickets:
                svo.wait_gather(tickets[r])  # the list this render overwrites may still be read by the previous frame's gather
            svo.render_device(u, w, h, lists[r].data_ptr(), tile_rank=r, tile_count=count)
            tickets[r] = svo.gather_tiles(lists[r].data_ptr(), per * 32 * 32 * 16, gathered[r].data_ptr(), root=0)
        svo.assemble_tiles(gathered.data_ptr(), per * 32 * 32 * 4, count, w, h, out.data_ptr(), stream=svo.comm_stream)
    svo.sync()
    assert out.cpu().numpy().tobytes() == full.tobytes()
    svo.comm_destroy()
    svo.comm_destroy()
