import os, sys, json
os.environ["VX_TIMELINE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from _pkg import load_package
vra = load_package()
from voxel_rs_amd import hip, scenes
import numpy as np, torch
fmtname, depth = sys.argv[1], int(sys.argv[2])
fmt = vra.SVO_ESVO if fmtname == "esvo" else vra.SVO_CSVO
W, H = 3840, 2160
world = vra.World(fmt); st = world.build_heightfield(depth)
svo = hip.Svo(fmt, world.size_in_bytes + (16 << 20))
svo.set_materials(scenes.synthetic_materials()); svo.set_textures(scenes.synthetic_textures(), 6); svo.update(world)
svo.set_frames_in_flight(1)
u = scenes.bench_camera(depth, st["h_max"], W, H, shadow_distance=3.0e38, render_shadows=True)
image = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
for _ in range(3): svo.render_device(u, W, H, image.data_ptr())
svo.sync()
t = svo.timeline()
flat = t.reshape(-1)
for wv in range(2):
    words = flat[32768 + wv * 2048: 32768 + (wv + 1) * 2048]
    words = words[words != 0]
    print(fmtname, depth, "wave", 1000 + wv, "loop calls", len(words), "trips", int((words & 0xffff).sum()))
    print("  trips | traversing at entry (shadow) -> at exit | foreign leaf idle")
    for w in words[:70]:
        w = int(w)
        print(f"  {w & 0xffff:4d} | {(w >> 16) & 0xff:2d} ({(w >> 24) & 0xff:2d}) -> {(w >> 32) & 0xff:2d} | {(w >> 40) & 0xff:2d} {(w >> 48) & 0xff:2d} {(w >> 56) & 0xff:2d}")
