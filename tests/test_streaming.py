"""World streaming: chunk loader -> generated chunks at their LOD -> SVO leaves -> dirty ranges ->
vx_commit, with frames rendered in between. CPU: the streamed world is the terrain it should be. GPU: the incrementally
committed device buffer renders exactly what the oracle renders from a full serialization of the same world."""
import math

import numpy as np
import pytest

from helpers import orc, vra  # noqa: F401
from voxel_rs_amd import host, scenes

SCENE_DEPTH = 9   # 512^3 domain, 16 chunks per axis
SEED = 0x5EED0001


def height(x, z):
    return int(host.lib().vxh_scene_height(SCENE_DEPTH, SEED, x, z))


@pytest.mark.parametrize("svo_type", [host.SVO_ESVO, host.SVO_CSVO])
def test_streamed_world_is_the_terrain(svo_type):
    radius = 5
    s = host.WorldStreamer(svo_type, SCENE_DEPTH, radius, 0, 8, SEED)
    eye = (200.5, 60.0, 230.5)
    n_events = s.move_to(*eye)
    assert n_events > 100
    applied = 0
    while True:
        st = s.pump(None, 400)  # dry run: no device
        applied += st["events"]
        if st["pending"] == 0:
            break
    assert applied == n_events and s.resident_chunks > 20
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    scene = orc.OracleScene(svo_type, s.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    rng = np.random.default_rng(1)
    checked = 0
    for _ in range(300):
        wx, wz = float(rng.uniform(eye[0] - 100, eye[0] + 100)), float(rng.uniform(eye[2] - 100, eye[2] + 100))
        if math.hypot(wx - eye[0], wz - eye[2]) > 32.0 * (radius - 1.5):
            continue
        top = s.to_svo((wx, 250.0, wz))
        r, _, _ = scene.intersect(top, (0.0, -1.0, 0.0), -1.0, False)
        assert r.t > 0, (wx, wz)
        # every chunk this close is LOD 5 (full detail): the ray lands on top of the terrain column
        surface_svo_y = top[1] - r.t
        surface_world_y = 250.0 - r.t
        assert abs(surface_world_y - (height(int(wx), int(wz)) + 1)) < 1e-3, (wx, wz, surface_world_y, surface_svo_y)
        checked += 1
    assert checked > 100


@pytest.mark.gpu
@pytest.mark.parametrize("first_image_bytes", [None, 8192], ids=["image-buffer-32MB", "image-buffer-8KB-doubling"])
@pytest.mark.parametrize("svo_type", [host.SVO_ESVO, host.SVO_CSVO])
def test_incremental_commits_render_like_a_full_upload(svo_type, first_image_bytes, monkeypatch):
    """... and with a first image buffer of 8 KB the image outgrows its device buffer a dozen times on the way: what it holds is carried over on the
    device into a buffer twice the size, only the commit's own ranges travel (runtime.cpp commit_now)."""
    from voxel_rs_amd import hip

    radius = 9  # LOD 5 within 6 chunks, LOD 4 beyond: both kinds resident, LOD changes while moving
    s = host.WorldStreamer(svo_type, SCENE_DEPTH, radius, 0, 8, SEED)
    if first_image_bytes:
        monkeypatch.setenv("VX_IMAGE_FIRST_BYTES", str(first_image_bytes))  # (read by vx_create)
    svo = hip.Svo(svo_type, 64 << 20)
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    svo.set_materials(mats)
    svo.set_textures(tex, 6)
    w, h = 160, 96
    path = [(150.5, 70.0, 150.5), (190.5, 66.0, 150.5), (260.5, 64.0, 200.5), (150.5, 70.0, 150.5)]  # ... and back: freed ranges are reused
    seen = dict(loads=0, unloads=0, lod_changes=0, bytes=0, frames=0)
    arena_after_first_visit = None
    for eye in path:
        s.move_to(*eye)
        while True:
            st = s.pump(svo._h, 120)  # several commits per move, a frame after each
            for k in ("loads", "unloads", "lod_changes", "bytes"):
                seen[k] += st[k]
            cam = s.to_svo(eye)
            u = scenes.render_params_to_uniforms(cam, (0.6, -0.45, 0.7), (0.0, 1.0, 0.0), math.radians(72.0), w / h, 0.3, (-1.0, -1.0, -1.0), True, 500.0)
            img, hits = svo.render(u, w, h, want_hits=True)
            scene = orc.OracleScene(svo_type, s.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
            cimg, chits = scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), w, h)
            assert hits.tobytes() == chits.tobytes(), f"hit records differ at {eye} with {st['pending']} events pending"
            assert np.nanmax(np.abs(img - cimg)) <= 5e-6
            seen["frames"] += 1
            if st["pending"] == 0:
                break
        if arena_after_first_visit is None:
            arena_after_first_visit = st["arena_bytes"]
    assert seen["loads"] > 500 and seen["unloads"] > 100 and seen["lod_changes"] > 50 and seen["frames"] >= 8
    assert (hits["flags"] & 1).mean() > 0.15
    # back at the start the same chunks are resident: the free list was reused instead of growing the arena without bound
    assert st["arena_bytes"] <= 1.5 * arena_after_first_visit
    assert svo.image_info()["image_bytes"] > 1 << 20  # (the image did outgrow an 8 KB buffer many times over)
