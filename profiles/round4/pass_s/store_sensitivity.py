#!/usr/bin/env python3
"""Does a frame's time depend on what its pixels are stored as? The C3 bench view, two frames in flight, into an RGBA32F and into an RGBA8 device image
(16 against 4 bytes a pixel), the still view 200 times each."""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
from _pkg import load_package  # noqa: E402

vra = load_package()
from voxel_rs_amd import hip, scenes  # noqa: E402


def main():
    import torch
    fmt = vra.SVO_CSVO if (len(sys.argv) < 2 or sys.argv[1] == "csvo") else vra.SVO_ESVO
    W, H, depth = 1920, 1080, 12
    world = vra.World(fmt)
    st = world.build_heightfield(depth)
    svo = hip.Svo(fmt, world.size_in_bytes + (16 << 20))
    svo.set_materials(scenes.synthetic_materials())
    svo.set_textures(scenes.asset_textures(ROOT / "tests" / "golden" / "textures"), 6)
    svo.update(world)
    u = scenes.bench_camera(depth, st["h_max"], W, H, shadow_distance=3.0e38, render_shadows=True)
    for name, f, dt in (("rgba32f", hip.VX_FORMAT_RGBA32F, torch.float32), ("rgba8", hip.VX_FORMAT_RGBA8, torch.uint8), ("rgba32f", hip.VX_FORMAT_RGBA32F, torch.float32), ("rgba8", hip.VX_FORMAT_RGBA8, torch.uint8)):
        img = [torch.zeros((H, W, 4), dtype=dt, device="cuda") for _ in range(2)]
        torch.cuda.synchronize()
        for k in range(20):
            svo.render_device(u, W, H, img[k & 1].data_ptr(), fmt=f)
        svo.sync()
        best = 1e9
        for rep in range(5):
            t0 = time.perf_counter()
            for k in range(200):
                svo.render_device(u, W, H, img[k & 1].data_ptr(), fmt=f)
            svo.sync()
            best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
        print(f"{sys.argv[1] if len(sys.argv) > 1 else 'csvo'} {name}: {best:.4f} ms a frame (still view, two in flight)")


if __name__ == "__main__":
    main()
