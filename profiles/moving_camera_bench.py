#!/usr/bin/env python3
"""C3 with the camera in motion: bench.py renders one view over and over, which lets the 256 MB infinity cache keep the frame's
octants from one frame to the next. Here the eye flies over the terrain and turns a little every frame (a new view matrix per
frame, 240 distinct frames), rays counted per view by the instrumented kernel beforehand.

    python profiles/moving_camera_bench.py --format csvo
"""
import argparse
import json
import math
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from _pkg import load_package  # noqa: E402

vra = load_package()
from voxel_rs_amd import hip, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--format", default="csvo")
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=240)
    ap.add_argument("--speed", type=float, default=4.0, help="blocks per frame (240 blocks/s at 60 Hz)")
    args = ap.parse_args()
    import torch

    fmt = vra.SVO_ESVO if args.format == "esvo" else vra.SVO_CSVO
    world = vra.World(fmt)
    st = world.build_heightfield(args.depth)
    svo = hip.Svo(fmt, world.size_in_bytes + (16 << 20))
    svo.set_materials(scenes.synthetic_materials())
    svo.set_textures(scenes.synthetic_textures(), 6)
    svo.update_full(world)
    W, H = args.width, args.height
    n = float(1 << args.depth)
    views = []
    for f in range(args.frames):
        a = 0.004 * f  # ~14 degrees per second at 60 Hz
        eye = (0.3 * n + args.speed * f * 0.6, st["h_max"] + 0.05 * n, 0.3 * n + args.speed * f * 0.7)
        fwd = (0.6 * math.cos(a) - 0.7 * math.sin(a), -0.35, 0.6 * math.sin(a) + 0.7 * math.cos(a))
        views.append(scenes.render_params_to_uniforms(eye, fwd, (0.0, 1.0, 0.0), math.radians(72.0), W / H, 0.3, (-1.0, -1.0, -1.0), True, 3.0e38))
    rays = sum(svo.render_counters(u, W, H)["rays"] for u in views[::8]) * 8  # every 8th view counted
    images = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    out = {"workload": f"C3 geometry ({W}x{H}, depth {args.depth}, {args.format.upper()}), {args.frames} distinct views: {args.speed} blocks and 0.23 degrees per frame"}
    for name, seq in (("moving", views), ("static (the first view repeated)", [views[0]] * args.frames)):
        for u in seq[:10]:
            svo.render_device(u, W, H, images[0].data_ptr())
        svo.sync()
        t0 = time.perf_counter()
        for i, u in enumerate(seq):
            svo.render_device(u, W, H, images[i % 2].data_ptr())
        svo.sync()
        ms = (time.perf_counter() - t0) * 1e3 / len(seq)
        r = rays if name == "moving" else svo.render_counters(views[0], W, H)["rays"] * args.frames
        out[name] = {"ms_per_frame": round(ms, 4), "Mrays_per_s": round(r / len(seq) / ms / 1e3, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
