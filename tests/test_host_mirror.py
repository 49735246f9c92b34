"""Host mirror above the C ABI: picker batches (known-answer tables of src/graphics/svo_picker.rs:311-536) on the CPU;
graphics::Svo and worldsvo::Svo end to end on the GPU."""
from pathlib import Path

import numpy as np
import pytest

from helpers import vra  # noqa: F401
from voxel_rs_amd import host

GOLD = Path(__file__).resolve().parent / "golden"


def test_picker_batch_serialization(golden):
    g = golden["picker_batch"]["serialization"]
    tasks = host.picker_serialize(g["rays"], g["aabbs"])
    # [2 rays] + [unit AABB: 3 rays x 8 corners] + [1.5^3 AABB: 24 + 24 + 6] = 80 (svo_picker.rs:330-332)
    assert len(tasks) == 80 == len(g["expected_tasks"])
    for got, exp in zip(tasks, g["expected_tasks"]):
        assert float(got["max_dst"]) == exp["max_dst"]
        np.testing.assert_array_equal(got["pos"], np.float32(exp["pos"]))
        np.testing.assert_array_equal(got["dir"], np.float32(exp["dir"]))


def test_picker_batch_deserialization(golden):
    g = golden["picker_batch"]["deserialization"]
    res = np.zeros(len(g["results"]), dtype=host.PICKER_RESULT_DTYPE)
    for i, r in enumerate(g["results"]):
        res[i]["dst"], res[i]["inside_voxel"], res[i]["pos"], res[i]["normal"] = r["dst"], int(r["inside_voxel"]), r["pos"], r["normal"]
    rays, aabbs = host.picker_deserialize(g["rays"], g["aabbs"], res)
    for got, exp in zip(rays, g["expected_rays"]):
        assert got[0] == exp["dst"] and bool(got[1]) == exp["inside_voxel"]
        np.testing.assert_array_equal(got[2:5], np.float32(exp["pos"]))
        np.testing.assert_array_equal(got[5:8], np.float32(exp["normal"]))
    for got, exp in zip(aabbs, g["expected_aabbs"]):
        np.testing.assert_array_equal(got[0:3], np.float32(exp["neg"]))
        np.testing.assert_array_equal(got[3:6], np.float32(exp["pos"]))


@pytest.mark.gpu
@pytest.mark.parametrize("svo_type", [host.SVO_ESVO, host.SVO_CSVO])
def test_graphics_svo_render_against_reference_png(svo_type, tmp_path):
    """src/graphics/svo.rs:342-399 end to end in C++: VoxelRegistry (PNG files) -> Svo::new -> update -> render -> as_image."""
    diff = host.reference_render_test(svo_type, GOLD / "textures", GOLD / "graphics_svo_render_expected.png", tmp_path / "actual.png")
    print("diff fraction", diff)
    assert diff < 0.001  # the reference's default threshold (svo.rs:393)


@pytest.mark.gpu
@pytest.mark.parametrize("svo_type", [host.SVO_ESVO, host.SVO_CSVO])
def test_worldsvo_mapper_world_space_raycasts(svo_type):
    """worldsvo::Svo: chunks keyed by WORLD chunk position, rays given and answered in world space, before and after the
    centre moves by one chunk (chunk shifting, worldsvo.rs:161-196). Floor heights differ per chunk so a wrong shift shows."""
    chunks = [(-1, 0, 0, 3), (0, 0, 0, 5), (1, 0, 0, 7), (0, 0, 1, 9), (2, 0, 0, 11)]
    xz = [(-16.5, 10.5), (5.5, 7.5), (40.5, 3.5), (8.5, 40.5), (70.5, 5.5)]
    first, second = host.mapper_raycast_test(svo_type, 2, chunks, [(0, 0, 0), (1, 0, 0)], xz, 30.0)
    tops = [3.0, 5.0, 7.0, 9.0, 11.0]
    for i, top in enumerate(tops):
        assert abs(first[i, 1] - top) < 1e-3 and abs(first[i, 0] - (30.0 - top)) < 1e-3, (i, first[i])
    # centre (1,0,0), radius 2: chunk (-1,0,0) is now 2 away on x (still inside), (0,0,1) is at distance sqrt(2) -> all still present
    for i, top in enumerate(tops):
        assert abs(second[i, 1] - top) < 1e-3 and abs(second[i, 0] - (30.0 - top)) < 1e-3, (i, second[i])


@pytest.mark.gpu
@pytest.mark.parametrize("svo_type", [host.SVO_ESVO, host.SVO_CSVO])
def test_physics_entities_settle_on_terrain(svo_type):
    """systems::Physics (src/systems/physics.rs) over the GPU picker: 48 entities dropped over a heightfield, stepped at the
    reference's 250 Hz with one picker launch per step for all of them, against the same steps with the oracle standing in
    as the Raycaster. Entity states must agree bit for bit after every step (every picker float is bit-exact)."""
    from helpers import orc
    from voxel_rs_amd import hip, scenes

    depth = 7
    world = vra.World(svo_type)
    st = world.build_heightfield(depth, threads=4)
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    scene = orc.OracleScene(svo_type, world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    svo = hip.Svo(svo_type, world.size_in_bytes + (1 << 20))
    svo.set_materials(mats)
    svo.set_textures(tex, 6)
    svo.update(world)

    rng = np.random.default_rng(3)
    n = 48
    pos = np.stack([rng.uniform(8, 120, n), st["h_max"] + rng.uniform(1.0, 6.0, n), rng.uniform(8, 120, n)], axis=1).astype(np.float32)
    gpu_e = host.make_entities(pos)
    gpu_e[:, 3] = rng.uniform(-6, 6, n)  # horizontal velocities: walls matter too
    gpu_e[:, 5] = rng.uniform(-6, 6, n)
    gpu_e[::7, 12] = 1.0                 # a few with wall_clip
    cpu_e = gpu_e.copy()
    dt = np.float32(1.0 / 250.0)
    total_tasks = 0
    steps = 300
    for step in range(steps):
        total_tasks += host.physics_step_many(svo._h, dt, 1, gpu_e)
        # the same step with the oracle as the Raycaster: batch -> tasks -> oracle casts -> AabbResults -> update_entity
        aabbs = [dict(pos=e[0:3], offset=e[6:9], extents=e[9:12]) for e in cpu_e]
        tasks = host.picker_serialize([], aabbs)
        res = np.zeros(len(tasks), dtype=host.PICKER_RESULT_DTYPE)
        for i, t in enumerate(tasks):
            r, _, _ = scene.intersect(t["pos"], t["dir"], float(t["max_dst"]), False)
            if r.t > 0:
                res[i]["dst"], res[i]["inside_voxel"], res[i]["pos"] = r.t, r.inside_voxel, list(r.pos)
                res[i]["normal"] = [[-1, 0, 0], [1, 0, 0], [0, -1, 0], [0, 1, 0], [0, 0, -1], [0, 0, 1]][r.face_id]
            else:
                res[i]["dst"] = -1
        _, aabb_results = host.picker_deserialize([], aabbs, res)
        host.physics_update(dt, cpu_e, aabb_results)
        assert gpu_e.tobytes() == cpu_e.tobytes(), f"entity states diverge at step {step}"
    assert total_tasks > steps * n * 20
    grounded = gpu_e[:, 16] == 1.0
    assert grounded.sum() > n // 2  # most have landed (1.2 s of free fall is 43 blocks; the terrain is at most 25 high)
    assert (gpu_e[:, 1] > 0.9).all() and (gpu_e[grounded, 4] == 0.0).all()
