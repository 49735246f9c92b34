#!/usr/bin/env python3
"""A deep configuration's frames and nothing else: the world built once, then for every setting a context of its own (the library reads its
environment knobs in vx_create), warm-up frames, `--frames` timed frames of a still view, wall clock per frame.

    python profiles/round6/deep_frames.py --format csvo --depth 14 --size 3840x2160 --frames 10 [--sweep "16:1 16:2 12:2 8:2"]

--sweep: "waves_per_cu:frames_in_flight" pairs (VX_WAVES_PER_CU is a knob of the measurement build: VX_LIB_DIR=voxel-rs_amd/lib/lib_tl).
Without it: one context with the environment as it stands -- the form the rocprofv3 passes of c4_counters.sh run.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from _pkg import load_package  # noqa: E402

vra = load_package()
from voxel_rs_amd import hip, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--format", default="csvo")
    ap.add_argument("--depth", type=int, default=14)
    ap.add_argument("--size", default="3840x2160")
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sweep", default="")
    ap.add_argument("--no-shadows", action="store_true")
    args = ap.parse_args()
    import torch

    W, H = (int(v) for v in args.size.split("x"))
    fmt = vra.SVO_ESVO if args.format == "esvo" else vra.SVO_CSVO
    t0 = time.perf_counter()
    world = vra.World(fmt)
    st = world.build_heightfield(args.depth)
    build_s = time.perf_counter() - t0
    u = scenes.bench_camera(args.depth, st["h_max"], W, H, shadow_distance=3.0e38, render_shadows=not args.no_shadows)
    out = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(4)]
    settings = [tuple(int(v) for v in s.split(":")) for s in args.sweep.split()] or [(0, 0)]
    for waves, fif in settings:
        if waves:
            os.environ["VX_WAVES_PER_CU"] = str(waves)
        if fif:
            os.environ["VX_FRAMES_IN_FLIGHT"] = str(fif)
        svo = hip.Svo(fmt, world.size_in_bytes + (16 << 20))
        svo.set_materials(scenes.synthetic_materials())
        svo.set_textures(scenes.synthetic_textures(), 6)
        t0 = time.perf_counter()
        svo.update_full(world)  # (a fresh context each time: the world's dirty ranges were consumed by the first update)
        commit_s = time.perf_counter() - t0
        for i in range(args.warmup):
            svo.render_device(u, W, H, out[i % 4].data_ptr())
        svo.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.frames):
            svo.render_device(u, W, H, out[i % 4].data_ptr())
        svo.sync()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / args.frames
        print(json.dumps({"format": args.format, "depth": args.depth, "size": args.size, "shadows": not args.no_shadows, "waves_per_cu": waves or "default",
                          "frames_in_flight": fif or os.environ.get("VX_FRAMES_IN_FLIGHT", "default"), "ms_per_frame": round(ms, 4),
                          "build_s": round(build_s, 2), "commit_s": round(commit_s, 2), "image": svo.image_info()}), flush=True)
        svo.close()


if __name__ == "__main__":
    main()
