"""Diagnostic: the five-waves build's frames against the frame with hit records, many contexts and frames."""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from helpers import SVO_TYPES, vra
from voxel_rs_amd import scenes, hip
fmt = sys.argv[1]; env = dict(kv.split("=") for kv in sys.argv[2].split(",")) if len(sys.argv) > 2 and sys.argv[2] else {}
os.environ.update(env)
world = vra.World(SVO_TYPES[fmt]); st = world.build_heightfield(8, threads=4)
tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
w, h = 250, 130
u = scenes.bench_camera(8, st["h_max"], w, h)
bad_frames = 0; total = 0
for rep in range(40):
    svo = hip.Svo(SVO_TYPES[fmt], world.size_in_bytes + (1 << 20)); svo.set_materials(mats); svo.set_textures(tex, 6); svo.update_full(world)
    img, hits = svo.render(u, w, h, want_hits=True)
    for f in range(8):
        img2, _ = svo.render(u, w, h)
        bad = np.argwhere((img2.view(np.uint32) != img.view(np.uint32)).any(axis=2))
        total += 1
        if len(bad):
            bad_frames += 1
            y, x = bad[0]
            if bad_frames <= 6: print(env, "rep", rep, "frame", f, "differing pixels", len(bad), bad[:4].tolist(), img2[y, x], img[y, x], hits[y, x], flush=True)
    svo.close()
print(fmt, env, os.environ.get("VX_LIB_DIR"), "bad frames", bad_frames, "of", total, flush=True)
