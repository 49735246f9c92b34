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
    args_hdr = Path(ROOT) / "voxel-rs_amd" / "csrc" / "hip" / "vx_args.hpp"
    shim = Path(ROOT) / "tests" / "cpp" / "shims" / "vx_platform.hpp"
    if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime, args_hdr.stat().st_mtime, shim.stat().st_mtime):
        # tests/cpp/shims comes first: its vx_platform.hpp (plain C++) is found instead of the product's (gfx950 built-ins)
        cmd = ["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-ffp-contract=off", "-mfma", f"-I{ROOT}/include", f"-I{ROOT}/tests/cpp/shims",
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
    frame = np.concatenate([frame, np.zeros(4, dtype=np.uint32)])  # the 16 zero bytes a context keeps behind the world buffer, inside its range (kWorldPad)
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


# ---- the traversal IMAGE of a world, walked by the device code the way a lane of the render kernel walks it ---------------------


RESULT_DTYPE = np.dtype([("t", "<f4"), ("value", "<u4"), ("face_id", "<i4"), ("pos", "<f4", 3), ("uv", "<f4", 2), ("color", "<f4", 4), ("lod", "<f4"),
                         ("inside_voxel", "<i4")])


def image_cast(lib, fmt, world, mats, tex, mips, tasks, cast_translucent, layout, shallow, walk_mode=2):
    from voxel_rs_amd import hip

    frame = world.frame(pad_words=0)
    head = 4 + (20 if fmt == "esvo" else 4)
    image, origin = hip.traversal_image(1 if fmt == "esvo" else 2, frame, frame.size * 4 - head, layout=layout, with_origin=True)
    image = np.concatenate([image, np.zeros(16, dtype=np.uint32)])  # the zero pad the context keeps behind the image
    origin = image  # (a CSVO world's origins are units of the image itself)
    levels = orc.mip_chain(tex, mips)
    chain = np.concatenate([lv.ravel() for lv in levels])
    offsets = np.cumsum([0] + [lv.size for lv in levels[:-1]])
    level_offset = (C.c_uint32 * 16)(*[int(o) for o in offsets])
    out = np.zeros(len(tasks), dtype=RESULT_DTYPE)
    steps = np.zeros(len(tasks), dtype=np.uint32)
    frame = np.concatenate([frame, np.zeros(4, dtype=np.uint32)])  # (kWorldPad, as above)
    lib.devhost_image_cast(1 if fmt == "esvo" else 2, layout, int(shallow), int(walk_mode), frame.ctypes.data_as(C.c_void_p), C.c_uint64(frame.size * 4),
                           image.ctypes.data_as(C.c_void_p), C.c_uint64(image.size * 4), origin.ctypes.data_as(C.c_void_p),
                           mats.ctypes.data_as(C.c_void_p), mats.size, chain.ctypes.data_as(C.c_void_p), tex.shape[2], tex.shape[1], tex.shape[0],
                           len(levels), level_offset, tasks.ctypes.data_as(C.c_void_p), len(tasks), int(cast_translucent),
                           out.ctypes.data_as(C.c_void_p), steps.ctypes.data_as(C.c_void_p))
    return out, steps


def oracle_cast(scene, tasks, cast_translucent):
    exp = np.zeros(len(tasks), dtype=RESULT_DTYPE)
    steps = np.zeros(len(tasks), dtype=np.uint32)
    for i, t in enumerate(tasks):
        ctr = orc.Counters()
        r, _, _ = scene.intersect(t["pos"], t["dir"], float(t["max_dst"]), bool(cast_translucent), counters=ctr)
        exp[i] = (r.t, r.value, r.face_id, list(r.pos), list(r.uv), list(r.color), r.lod, r.inside_voxel)
        steps[i] = ctr.iterations
    return exp, steps


def assert_same_casts(got, gsteps, exp, esteps):
    """Hit identity, every traversal float and the iteration count exactly; the sampled colour to 5e-6 (a trilinear sample blends
    its texels in byte units on the device, vx_device.hpp: sample_linear_bytes)."""
    exact = [n for n in RESULT_DTYPE.names if n != "color"]
    bad = [i for i in range(len(exp)) if any(got[i][n].tobytes() != exp[i][n].tobytes() for n in exact) or gsteps[i] != esteps[i]
           or np.abs(got[i]["color"] - exp[i]["color"]).max() > 5e-6]
    assert not bad, (len(bad), bad[:5], got[bad[0]], exp[bad[0]], gsteps[bad[0]], esteps[bad[0]])


@pytest.mark.parametrize("cast_translucent", [0, 1])
@pytest.mark.parametrize("layout", [1, 2])
@pytest.mark.parametrize("fmt", ["esvo", "csvo"])
def test_image_traversal_with_rays_from_inside_voxels(devhost, golden, fmt, layout, cast_translucent):
    """A dense random chunk: about one ray in five starts INSIDE a voxel and is led into it (svo.esvo.glsl:183-185,
    svo.csvo.glsl:293-295). On the image of an ESVO world that is a walk through an empty node; on the image of a CSVO world the
    ray makes an excursion onto the world's own bytes (phantom leaves included) and comes back. Results and iteration counts
    are the oracle's on the world's own bytes."""
    rng = np.random.default_rng(7)
    pts = rng.integers(0, 32, size=(6000, 3))
    blocks = [[int(x), int(y), int(z), int(rng.choice([1, 2, 3, 4]))] for x, y, z in pts]
    svo_pos = (3, 2, 1)
    scene, world = oracle_scene(golden, fmt, svo_pos, blocks)
    tex, mips = golden_textures(golden)
    mats = golden_materials(golden)
    base = np.asarray(svo_pos, dtype=np.float32) * 32
    tasks = random_tasks(rng, 1500, 0, 32)
    tasks["pos"] += base
    exp, esteps = oracle_cast(scene, tasks, cast_translucent)
    assert (exp["inside_voxel"] != 0).sum() > 100 and (exp["t"] > 0).sum() > 300
    for shallow in (True, False):
        got, gsteps = image_cast(devhost, fmt, world, mats, tex, mips, tasks, cast_translucent, layout, shallow)
        assert_same_casts(got, gsteps, exp, esteps)
        if fmt == "csvo":
            devhost.devhost_given_up.restype = C.c_uint32
            assert devhost.devhost_given_up() < 0.2 * (exp["inside_voxel"] != 0).sum()  # (the walk serves most of them itself)


def test_lean_walk_as_the_image_only_kernels_run_it(devhost, golden):
    """The render kernel's build of the walk inside a voxel: phantom leaves of opaque blocks are hits without their sample (the colour is
    sampled when the hit is shaded), any other phantom leaf is given up -- and what is given up is run whole on the world's own bytes:
    the same results and iteration counts. The rays are what the shadow rays of a deep world are: from a primary hit's position, which
    lies inside its voxel, towards the light."""
    from voxel_rs_amd import scenes

    world = vra.World(2)
    st = world.build_heightfield(9, threads=4)
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    scene = orc.OracleScene(2, world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    w, h = 96, 54
    u = scenes.bench_camera(9, st["h_max"], w, h, shadow_distance=3.0e38)
    _, hits = scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), w, h)
    sel = hits[(hits["flags"] & 2) != 0]
    tasks = np.zeros(len(sel), dtype=orc.PICKER_TASK_DTYPE)
    tasks["pos"] = sel["pos"]
    tasks["dir"] = -np.asarray(u.light_dir[:], dtype=np.float32)
    tasks["max_dst"] = -1
    exp, esteps = oracle_cast(scene, tasks, 1)
    assert (exp["inside_voxel"] != 0).sum() > 1000
    opaque = 0
    for b in range(len(mats)):
        if all(tex[max(0, int(mats[b][k]))][:, :, 3].min() > 0 for k in ("tex_top", "tex_side", "tex_bottom")):
            opaque |= 1 << b
    devhost.devhost_set_opaque(opaque & 0xffffffff, opaque >> 32)
    devhost.devhost_given_up.restype = C.c_uint32
    for walk_mode in (2, 3):
        for layout, shallow in ((1, True), (1, 2), (2, 2)):
            got, gsteps = image_cast(devhost, "csvo", world, mats.view(orc.MATERIAL_DTYPE), tex, 6, tasks, 1, layout, shallow, walk_mode=walk_mode)
            assert_same_casts(got, gsteps, exp, esteps)
            assert devhost.devhost_given_up() < 0.1 * len(tasks)


@pytest.mark.parametrize("fmt", ["esvo", "csvo"])
@pytest.mark.parametrize("base,depth", [((200, 3, 201), 13), ((400, 3, 401), 14), ((800, 3, 801), 15)])
def test_image_traversal_in_deep_worlds(devhost, golden, fmt, base, depth):
    """Chunks far from the origin of a deep world: voxels sit on or below the last LDS-resident stack level, so the walk into a
    voxel goes through the hand-over to the full stack (ESVO image at depth 14 and more) and the excursion returns to a node
    below the resident levels (CSVO image at depth 14 and more)."""
    rng = np.random.default_rng(11)
    world = vra.World(1 if fmt == "esvo" else 2)
    for dx in range(2):
        chunk = vra.Chunk(base[0] + dx, base[1], base[2], 5)
        for x in range(32):
            for z in range(32):
                for y in range(6 + int(rng.integers(0, 6))):
                    chunk.set_block(x, y, z, int(rng.choice([1, 2, 3, 4])))
        chunk.compact()
        world.set_chunk((base[0] + dx, base[1], base[2]), chunk)
    world.serialize()
    assert world.depth == depth
    tex, mips = golden_textures(golden)
    mats = golden_materials(golden)
    scene = orc.OracleScene(1 if fmt == "esvo" else 2, world.frame(), mats, tex, mips)
    tasks = random_tasks(rng, 1200, 0, 1)
    tasks["pos"] = (np.float32([32 * c for c in base]) + rng.uniform([0, 0, 0], [64, 14, 32], size=(len(tasks), 3))).astype(np.float32)
    tasks["max_dst"] = -1
    exp, esteps = oracle_cast(scene, tasks, 1)
    assert (exp["inside_voxel"] != 0).sum() > 100
    for layout in (1, 2):
        got, gsteps = image_cast(devhost, fmt, world, mats, tex, mips, tasks, 1, layout, shallow=False)
        assert_same_casts(got, gsteps, exp, esteps)
    if depth <= (13 if fmt == "esvo" else 14):  # what vx_render takes for "no push below the resident levels" (vx_api.hip, launch_render)
        got, gsteps = image_cast(devhost, fmt, world, mats, tex, mips, tasks, 1, 1, shallow=True)
        assert_same_casts(got, gsteps, exp, esteps)
    # the build with 16 resident stack levels (third plane 16 bits wide): every depth here without the hand-over
    for layout in (1, 2):
        got, gsteps = image_cast(devhost, fmt, world, mats, tex, mips, tasks, 1, layout, shallow=2)
        assert_same_casts(got, gsteps, exp, esteps)


def shadow_pairs(lib, fmt, world, mats, tex, mips, tasks, to_light):
    from voxel_rs_amd import hip

    frame = world.frame(pad_words=0)
    head = 4 + (20 if fmt == "esvo" else 4)
    image, origin = hip.traversal_image(1 if fmt == "esvo" else 2, frame, frame.size * 4 - head, layout=1, with_origin=True)
    image = np.concatenate([image, np.zeros(16, dtype=np.uint32)])
    origin = image
    levels = orc.mip_chain(tex, mips)
    chain = np.concatenate([lv.ravel() for lv in levels])
    offsets = np.cumsum([0] + [lv.size for lv in levels[:-1]])
    level_offset = (C.c_uint32 * 16)(*[int(o) for o in offsets])
    out = np.zeros(2 * len(tasks), dtype=RESULT_DTYPE)
    steps = np.zeros(2 * len(tasks), dtype=np.uint32)
    frame = np.concatenate([frame, np.zeros(4, dtype=np.uint32)])
    light = (C.c_float * 3)(*[float(v) for v in to_light])
    lib.devhost_image_shadow_pairs.restype = C.c_uint32
    taken = lib.devhost_image_shadow_pairs(1 if fmt == "esvo" else 2, frame.ctypes.data_as(C.c_void_p), C.c_uint64(frame.size * 4), image.ctypes.data_as(C.c_void_p),
                                           C.c_uint64(image.size * 4), origin.ctypes.data_as(C.c_void_p), mats.ctypes.data_as(C.c_void_p), mats.size,
                                           chain.ctypes.data_as(C.c_void_p), tex.shape[2], tex.shape[1], tex.shape[0], len(levels), level_offset,
                                           tasks.ctypes.data_as(C.c_void_p), len(tasks), light, out.ctypes.data_as(C.c_void_p), steps.ctypes.data_as(C.c_void_p))
    return taken, out, steps


@pytest.mark.parametrize("fmt", ["esvo", "csvo"])
@pytest.mark.parametrize("depth", [9, 11])
def test_shadow_rays_that_start_on_their_primary_rays_path(devhost, fmt, depth):
    """render_persistent's ray set-up lets a pixel's shadow ray take the levels down to the voxel its primary hit in one go, along the nodes the primary's stack
    holds (Trav::descend_along: an image cursor writes its parent's entry at every push). The device code, driven the way the kernel drives it: for the
    primary rays of a view of a terrain, the shadow ray (hit + normal * 0.001 towards the light, world.glsl:79-84) from the root like any ray and along the
    path -- both must be the ORACLE's shadow ray, results and iteration counts, and most of them must in fact have taken the path."""
    from voxel_rs_amd import scenes

    svo_type = 1 if fmt == "esvo" else 2
    world = vra.World(svo_type)
    st = world.build_heightfield(depth, threads=4)
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    scene = orc.OracleScene(svo_type, world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    w, h = 80, 45
    u = scenes.bench_camera(depth, st["h_max"], w, h, shadow_distance=3.0e38)
    eye = np.asarray(u.cam_pos[:], dtype=np.float32)
    # the view's primary rays as picker tasks: towards the hit positions of an oracle render (any rays towards the terrain would do)
    _, hits = scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), w, h)
    sel = hits[(hits["flags"] & 1) != 0]
    tasks = np.zeros(len(sel), dtype=orc.PICKER_TASK_DTYPE)
    d = sel["pos"] - eye
    tasks["pos"] = eye
    tasks["dir"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    tasks["max_dst"] = -1
    opaque = 0
    for b in range(len(mats)):
        if all(tex[max(0, int(mats[b][k]))][:, :, 3].min() > 0 for k in ("tex_top", "tex_side", "tex_bottom")):
            opaque |= 1 << b
    devhost.devhost_set_opaque(opaque & 0xffffffff, opaque >> 32)
    to_light = -np.asarray(u.light_dir[:], dtype=np.float32)
    taken, out, steps = shadow_pairs(devhost, fmt, world, mats.view(orc.MATERIAL_DTYPE), tex, 6, tasks, to_light)
    # the oracle's shadow rays: from ITS primary hits (the device's primaries are the oracle's: test_image_traversal_*)
    prim, _ = oracle_cast(scene, tasks, 1)
    normals = np.asarray(NORMALS, dtype=np.float32)
    stasks = np.zeros(len(tasks), dtype=orc.PICKER_TASK_DTYPE)
    stasks["pos"] = prim["pos"] + normals[np.clip(prim["face_id"], 0, 5)] * np.float32(0.001)
    stasks["dir"] = to_light
    stasks["max_dst"] = -1
    exp, esteps = oracle_cast(scene, stasks, 1)
    hit = prim["t"] >= 0
    assert hit.sum() > 1000 and taken > 0.9 * hit.sum(), (int(hit.sum()), taken)
    for k in (0, 1):  # from the root / along the path
        got, gsteps = out[k::2][hit], steps[k::2][hit]
        assert_same_casts(got, gsteps, exp[hit], esteps[hit])


@pytest.mark.parametrize("fmt", ["esvo", "csvo"])
def test_shadow_rays_on_the_path_in_a_deep_world(devhost, golden, fmt):
    """... and at depth 14, where the shadow ray's offset falls below the clamp that keeps a hit inside its voxel: every shadow ray starts INSIDE the voxel its
    primary hit (the result says so for those whose t_min is 0 there), takes the whole path down to the voxel's parent in one go and is led into the voxel from there (an empty node of an ESVO world; the walk on
    the world's own bytes of a CSVO world)."""
    rng = np.random.default_rng(5)
    base = (400, 3, 401)
    svo_type = 1 if fmt == "esvo" else 2
    world = vra.World(svo_type)
    for dx in range(2):
        chunk = vra.Chunk(base[0] + dx, base[1], base[2], 5)
        for x in range(32):
            for z in range(32):
                for y in range(6 + int(rng.integers(0, 6))):
                    chunk.set_block(x, y, z, int(rng.choice([1, 2, 3, 4])))
        chunk.compact()
        world.set_chunk((base[0] + dx, base[1], base[2]), chunk)
    world.serialize()
    assert world.depth == 14
    tex, mips = golden_textures(golden)
    mats = golden_materials(golden)
    scene = orc.OracleScene(svo_type, world.frame(), mats, tex, mips)
    tasks = random_tasks(rng, 1500, 0, 1)
    tasks["pos"] = (np.float32([32 * c for c in base]) + rng.uniform([0, 20, 0], [64, 30, 32], size=(len(tasks), 3))).astype(np.float32)
    d = rng.normal(size=(len(tasks), 3)) * [1.0, 0.2, 1.0] - [0, 1.0, 0]
    tasks["dir"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    tasks["max_dst"] = -1
    devhost.devhost_set_opaque(0, 0)
    to_light = np.float32([0.57735026, 0.57735026, 0.57735026])
    taken, out, steps = shadow_pairs(devhost, fmt, world, mats, tex, mips, tasks, to_light)
    prim, _ = oracle_cast(scene, tasks, 1)
    hit = (prim["t"] >= 0) & (prim["inside_voxel"] == 0)
    normals = np.asarray(NORMALS, dtype=np.float32)
    stasks = np.zeros(len(tasks), dtype=orc.PICKER_TASK_DTYPE)
    stasks["pos"] = prim["pos"] + normals[np.clip(prim["face_id"], 0, 5)] * np.float32(0.001)
    stasks["dir"] = to_light
    stasks["max_dst"] = -1
    exp, esteps = oracle_cast(scene, stasks, 1)
    assert hit.sum() > 500 and taken > 0.9 * hit.sum() and (exp["inside_voxel"][hit] != 0).mean() > 0.3, (int(hit.sum()), taken)
    for k in (0, 1):
        assert_same_casts(out[k::2][hit], steps[k::2][hit], exp[hit], esteps[hit])


def test_the_sub_tile_queue_hands_out_every_sub_tile_once(devhost):
    """queue_subtile (vx_args.hpp): the launch's sub-tiles are dealt out to the eight dispensers in stretches of `stripe`; a dispenser's numbers grow, the
    first beyond the launch means it is dry, and between them the dispensers cover the launch exactly once -- for launches of a tile, of odd sizes, with
    stretches of one sub-tile (the cost order), a tile, an eighth of the launch, more than the launch."""
    for total in (16, 32, 48, 16 * 7, 16 * 90, 16 * 2040, 16 * 8160):
        for stripe in (1, 2, 16, 48, 100, 960, (total + 7) // 8, total, total + 5):
            assert devhost.devhost_queue_covers(total, stripe) == 1, (total, stripe)


def test_tile_numberings_are_permutations(devhost):
    """RenderParams::tile_numbering as launch_render sets it up (set_tile_numbering) and the kernel uses it both ways (tile_place for the refill, tile_number
    for the cost notes): along the rows, in strips of W columns (the last strip narrower), with the 0.618 stride (what a tile list gets whatever is asked
    for) -- each a permutation of the launch's tiles whose inverse is the inverse. The strips run down the screen: consecutive numbers of a strip are
    neighbours in a row, then the next row."""
    for tx, ty in ((1, 1), (2, 1), (1, 5), (3, 3), (10, 9), (60, 34), (120, 68), (7, 13)):
        n = tx * ty
        for numbering in (0, 1, 2):
            for strip in (1, 2, 3, 4, 8, 100):
                assert devhost.devhost_tile_numbering(tx, ty, 1, n, numbering, strip, None) == 1, (tx, ty, numbering, strip)
        for count in (2, 3, 8):  # a rank's share of a tile list
            for rank_n in {(n + count - 1) // count, n // count} - {0}:
                for numbering in (0, 1, 2):
                    assert devhost.devhost_tile_numbering(tx, ty, count, rank_n, numbering, 8, None) == 1, (tx, ty, count, rank_n, numbering)
    place = (C.c_uint32 * (60 * 34))()
    assert devhost.devhost_tile_numbering(60, 34, 1, 60 * 34, 1, 8, place) == 1
    p = np.frombuffer(place, dtype=np.uint32)
    assert list(p[:9]) == [0, 1, 2, 3, 4, 5, 6, 7, 60] and p[8 * 34] == 8  # a strip of eight columns from the top row down, then the next strip
    assert p[7 * 8 * 34] == 56 and list(p[7 * 8 * 34:7 * 8 * 34 + 5]) == [56, 57, 58, 59, 116]  # the last strip: four columns
    assert devhost.devhost_tile_numbering(60, 34, 1, 60 * 34, 2, 8, place) == 1
    step = int(p[1]) - int(p[0])
    assert 0.55 * 2040 < step < 0.68 * 2040  # the stride: about 0.618 of the tiles
