"""Diagnostic: test_kernel_versions_agree's scenario, env by env and frame by frame, with the number and place of differing pixels."""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from helpers import SVO_TYPES, vra
from voxel_rs_amd import scenes, hip

fmt = sys.argv[1] if len(sys.argv) > 1 else "esvo"
world = vra.World(SVO_TYPES[fmt]); st = world.build_heightfield(8, threads=4)
tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
w, h = 250, 130
u = scenes.bench_camera(8, st["h_max"], w, h)
envs = ({"VX_RENDER_KERNEL": "1"}, {"VX_RENDER_KERNEL": "2"}, {"VX_RENDER_KERNEL": "2", "VX_REFILL_MIN": "1", "VX_SERVICE_MIN": "1"},
        {"VX_RENDER_KERNEL": "2", "VX_REFILL_MIN": "64", "VX_SERVICE_MIN": "64"}, {"VX_RENDER_KERNEL": "2", "VX_REFILL_MIN": "7", "VX_SERVICE_MIN": "33"},
        {"VX_TRAVERSAL_IMAGE": "0"}, {"VX_WIDE_IMAGE": "1"}, {"VX_HOT_FIRST": "0"}, {"VX_FIVE_WAVES": "1"}, {"VX_BATCH": "1"},
        {"VX_WIDE_IMAGE": "1", "VX_MIN_WAVES": "1"}, {"VX_HOT_LEVELS": "1"}, {"VX_FOREIGN_MIN": "1"}, {"VX_TICKET_AHEAD": "1"},
        {"VX_TICKET_AHEAD": "0", "VX_FRAMES_IN_FLIGHT": "3"}, {"VX_FOREIGN_RERUN": "0"})
if len(sys.argv) > 2: envs = envs[int(sys.argv[2]):]
for env in envs:
    for k in list(os.environ):
        if k.startswith("VX_") and k != "VX_LIB_DIR": del os.environ[k]
    os.environ.update(env)
    svo = hip.Svo(SVO_TYPES[fmt], world.size_in_bytes + (1 << 20)); svo.set_materials(mats); svo.set_textures(tex, 6); svo.update_full(world)
    img, hits = svo.render(u, w, h, want_hits=True)
    for f in range(5):
        img2, _ = svo.render(u, w, h)
        d = (img2.view(np.uint32) != img.view(np.uint32)).any(axis=2)
        bad = np.argwhere(d)
        print(env, "frame", f, "differing pixels", len(bad), bad[:6].tolist(), flush=True)
        if len(bad):
            y, x = bad[0]; print("   got", img2[y, x], "ref", img[y, x])
    svo.close()
