#!/usr/bin/env python3
"""One timing line per GPU configuration of BASELINE.json on ONE MI355X (C3 is bench.py's headline; C4's streaming version is
profiles/stream_bench.py): frames rendered back to back into device memory, wall clock per frame, rays from the instrumented
kernel. C5's frame here is what ONE GPU would do alone: the 7680x4320 supersample render plus the 2x2 resolve.

    python profiles/configs_bench.py --format csvo [--configs C2 C3 C4 C5]
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from _pkg import load_package  # noqa: E402

vra = load_package()
from voxel_rs_amd import hip, scenes  # noqa: E402

CONFIGS = {
    # name: (depth, width, height, shadows, supersample)
    "C2": (10, 1920, 1080, False, 1),
    "C3": (12, 1920, 1080, True, 1),
    "C4": (14, 3840, 2160, True, 1),   # the static full-detail depth-14 terrain (CSVO 1.4 GB, ESVO 5 GB: a world buffer beyond 4 GiB)
    "C5": (14, 3840, 2160, True, 2),
    "C4-primary": (14, 3840, 2160, False, 1),
    "C4-d13": (13, 3840, 2160, True, 1),   # round 1's stand-in (then the deepest terrain a world buffer could hold)
    "C5-d13": (13, 3840, 2160, True, 2),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--format", default="csvo")
    ap.add_argument("--configs", nargs="+", default=["C2", "C3", "C4", "C5"])
    ap.add_argument("--steps", type=int, default=30)
    args = ap.parse_args()
    import torch

    fmt = vra.SVO_ESVO if args.format == "esvo" else vra.SVO_CSVO
    worlds = {}
    for name in args.configs:
        depth, w, h, shadows, ss = CONFIGS[name]
        if depth not in worlds:
            worlds.clear()  # (one world at a time: a depth-14 terrain takes gigabytes on both sides)
            t0 = time.perf_counter()
            world = vra.World(fmt)
            st = world.build_heightfield(depth)
            build_s = time.perf_counter() - t0
            svo = hip.Svo(fmt, world.size_in_bytes + (16 << 20))
            svo.set_materials(scenes.synthetic_materials())
            svo.set_textures(scenes.synthetic_textures(), 6)
            t0 = time.perf_counter()
            svo.update(world)
            print(json.dumps({"depth": depth, "format": args.format, "world_MB": round(world.size_in_bytes / 1e6, 1), "leaves": st["leaves"], "chunks": st["chunks"],
                              "build_s": round(build_s, 2), "commit_s": round(time.perf_counter() - t0, 2)}), flush=True)
            worlds[depth] = (world, st, svo)
        world, st, svo = worlds[depth]
        W, H = w * ss, h * ss
        u = scenes.bench_camera(depth, st["h_max"], W, H, shadow_distance=3.0e38, render_shadows=shadows)
        counters = svo.render_counters(u, W, H)
        rays = counters["rays"]
        big = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        small = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda") if ss > 1 else None
        torch.cuda.synchronize()
        stream = torch.cuda.current_stream().cuda_stream

        def frame(i):
            svo.render_device(u, W, H, big[i % 2].data_ptr())
            if ss > 1:
                svo.stream_wait_render(stream)
                svo.resolve_2x2(big[i % 2].data_ptr(), w, h, small.data_ptr(), stream=stream)

        for i in range(4):
            frame(i)
        svo.sync()
        torch.cuda.synchronize()
        # the walks inside voxels, counted over frames of their own: the counting costs frame time (vx_excursion_counters)
        svo.excursion_counters(reset=True)
        for i in range(args.steps):
            frame(i)
        svo.sync()
        torch.cuda.synchronize()
        exc = svo.excursion_counters(stop=True)
        t0 = time.perf_counter()
        for i in range(args.steps):
            frame(i)
        svo.sync()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / args.steps
        print(json.dumps({"config": name, "format": args.format, "depth": depth, "width": w, "height": h, "supersample": ss, "shadows": shadows,
                          "world_MB": round(world.size_in_bytes / 1e6, 1), "rays_per_frame": int(rays), "ms_per_frame": round(ms, 4),
                          "Mrays_per_s": round(rays / ms / 1e3, 1),
                          # work per ray grows with the depth of the tree: iterations of the traversal loop per second compare across configurations
                          "iterations_per_ray": round(counters["iterations"] / rays, 1), "Giterations_per_s": round(counters["iterations"] / ms / 1e6, 2),
                          "rays_led_into_a_voxel_per_frame": exc["rays"] // args.steps, "of_which_started_over": exc["started_over"] // args.steps,
                          "excursion_phases_per_frame": exc["service_phases"] // args.steps, "iterations_on_bytes_per_frame": exc["iterations_on_bytes"] // args.steps,
                          "image": svo.image_info()}), flush=True)


if __name__ == "__main__":
    main()
