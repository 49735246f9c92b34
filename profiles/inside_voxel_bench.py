#!/usr/bin/env python3
"""The camera sits INSIDE a voxel, so every primary ray is one the traversal image cannot serve. The renderer notices (a point
query on the host mirror of the image) and sends such frames straight to the kernel that traverses the world's own bytes;
with VX_EYE_CHECK=0 (set by the caller) they go through the image kernel instead, where every pixel ends up in the second
phase. Compared with a camera just above the same spot."""
import json
import math
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from _pkg import load_package  # noqa: E402

vra = load_package()
from voxel_rs_amd import hip, host, scenes  # noqa: E402


def main():
    import torch

    depth = 12
    world = vra.World(vra.SVO_CSVO)
    st = world.build_heightfield(depth)
    svo = hip.Svo(vra.SVO_CSVO, world.size_in_bytes + (16 << 20))
    svo.set_materials(scenes.synthetic_materials())
    svo.set_textures(scenes.synthetic_textures(), 6)
    svo.update(world)
    W, H = 1920, 1080
    n = float(1 << depth)
    x, z = 0.5 * n + 0.3, 0.5 * n + 0.6
    ground = float(host.lib().vxh_scene_height(depth, 0x5EED0001, int(x), int(z)))
    images = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    out = {}
    for label, y in (("inside the top voxel", ground + 0.5), ("two blocks above it", ground + 3.0)):
        u = scenes.render_params_to_uniforms((x, y, z), (0.6, 0.15, 0.7), (0.0, 1.0, 0.0), math.radians(72.0), W / H, 0.3, (-1.0, -1.0, -1.0), True, 3.0e38)
        for i in range(6):
            svo.render_device(u, W, H, images[i % 2].data_ptr())
        svo.sync()
        t0 = time.perf_counter()
        steps = 40
        for i in range(steps):
            svo.render_device(u, W, H, images[i % 2].data_ptr())
        svo.sync()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        rays = svo.render_counters(u, W, H)["rays"]
        out[label] = {"ms_per_frame": round(ms, 3), "Mrays_s": round(rays / ms / 1e3, 1), "rays": int(rays)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
