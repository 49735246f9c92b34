#!/usr/bin/env python3
"""C3 with a camera that moves every frame (turns by --degrees per frame and walks forward): what "sorted passes" and "expensive sub-tiles first" --
both use earlier frames of the view -- are worth when the view is not the same twice. Device-resident frames, two in flight, like bench.py.

    python profiles/moving_camera.py --format esvo [--degrees 0.25] [--frames 400]
"""
import argparse, json, math, sys, time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from _pkg import load_package  # noqa: E402

vra = load_package()
from voxel_rs_amd import hip, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--format", default="esvo")
    ap.add_argument("--degrees", type=float, default=0.25)
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--start", type=float, default=0.0, help="the view's yaw at frame 0, degrees")
    ap.add_argument("--pitch", type=float, default=-0.35, help="the view direction's y component")
    args = ap.parse_args()
    import torch

    fmt = vra.SVO_ESVO if args.format == "esvo" else vra.SVO_CSVO
    W, H, depth = 1920, 1080, 12
    world = vra.World(fmt)
    st = world.build_heightfield(depth)
    svo = hip.Svo(fmt, world.size_in_bytes + (16 << 20))
    svo.set_materials(scenes.synthetic_materials())
    svo.set_textures(scenes.asset_textures(ROOT / "tests" / "golden" / "textures"), 6)
    svo.update(world)
    svo.set_frames_in_flight(2)
    n = float(1 << depth)
    images = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]

    def uniforms(i):
        a = math.radians(args.start + args.degrees * i)
        fwd = (0.6 * math.cos(a) - 0.7 * math.sin(a), args.pitch, 0.6 * math.sin(a) + 0.7 * math.cos(a))
        eye = (0.5 * n + 0.02 * i * (args.degrees != 0), st["h_max"] + 0.05 * n, 0.5 * n)
        return scenes.render_params_to_uniforms(eye, fwd, (0.0, 1.0, 0.0), math.radians(72.0), W / H, 0.3, (-1.0, -1.0, -1.0), True, 3.0e38)

    us = [uniforms(i) for i in range(args.frames)]
    for i in range(20):
        svo.render_device(us[0], W, H, images[i % 2].data_ptr())
    svo.sync()
    t0 = time.perf_counter()
    for i in range(args.frames):
        svo.render_device(us[i], W, H, images[i % 2].data_ptr())
    svo.sync()
    ms = (time.perf_counter() - t0) * 1e3 / args.frames
    print(json.dumps({"format": args.format, "degrees_per_frame": args.degrees, "start": args.start, "pitch": args.pitch, "frames": args.frames, "ms_per_frame": round(ms, 4)}))


if __name__ == "__main__":
    main()
