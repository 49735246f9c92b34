#!/usr/bin/env python3
"""A/B sweeps of render-kernel variants in ONE process (interleaved rounds, median and min of the HIP-event kernel time).

    python profiles/sweep.py --format esvo --configs "k=1" "k=2" "k=2,r=8,s=20" --rounds 3 --steps 10
"""
import argparse
import os
import statistics
import time
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
# (the knobs of experiments -- r, w, h, t, q -- are read by the library's measurement build only: make tl)
os.environ.setdefault("VX_LIB_DIR", str(ROOT / "voxel-rs_amd" / "lib" / "lib_tl"))
sys.path.insert(0, str(ROOT))
from _pkg import load_package  # noqa: E402

vra = load_package()
from voxel_rs_amd import hip, scenes  # noqa: E402

KEYS = {"k": "VX_RENDER_KERNEL", "r": "VX_REFILL_MIN", "s": "VX_SERVICE_MIN", "w": "VX_WAVES_PER_CU", "f": "VX_FRAMES_IN_FLIGHT", "h": "VX_HOT_FIRST", "X": "VX_HOT_LEVELS",
        "L": "VX_LIB_DIR", "R": "VX_FOREIGN_RERUN", "n": "VX_TILE_NUMBERING", "t": "VX_TILE_STRIP", "q": "VX_QUEUE_STRIPE", "i": "VX_TRAVERSAL_IMAGE"}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--format", default="esvo")
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--configs", nargs="+", required=True)
    ap.add_argument("--tiles", type=int, default=1, help="render only rank 0's share of an N-way tile sharding (what one of N GPUs does)")
    args = ap.parse_args()
    import torch

    fmt = vra.SVO_ESVO if args.format == "esvo" else vra.SVO_CSVO
    world = vra.World(fmt)
    st = world.build_heightfield(args.depth)
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    W, H = args.width, args.height
    u = scenes.bench_camera(args.depth, st["h_max"], W, H, shadow_distance=3.0e38)
    image = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    ctxs = []
    for cfg in args.configs:
        for k in KEYS.values():
            os.environ.pop(k, None)
        for kv in cfg.split(","):
            k, v = kv.split("=")
            os.environ[KEYS[k]] = v
        svo = hip.Svo(fmt, world.size_in_bytes + (16 << 20))
        svo.set_materials(mats)
        svo.set_textures(tex, 6)
        svo.update_full(world)
        for _ in range(3):
            svo.render_device(u, W, H, image.data_ptr(), tile_rank=0, tile_count=args.tiles)
        svo.sync()
        ctxs.append((cfg, svo, [], []))
    counters = ctxs[0][1].render_counters(u, W, H, 0, args.tiles)
    rays = counters["rays"]
    print("counters:", {k: int(v) for k, v in counters.items()})
    for _ in range(args.rounds):
        for cfg, svo, times, walls in ctxs:
            svo.profile_enable(True)
            svo.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                svo.render_device(u, W, H, image.data_ptr(), tile_rank=0, tile_count=args.tiles)
            svo.sync()
            walls.append((time.perf_counter() - t0) * 1e3 / args.steps)
            ms, n = svo.profile_read()
            svo.profile_enable(False)
            times.append(ms / n)
    for cfg, svo, times, walls in ctxs:
        med, mn, wall = statistics.median(times), min(times), statistics.median(walls)
        print(f"{args.format} {cfg:24s} kernel span ms median {med:.4f} min {mn:.4f} | wall ms/frame {wall:.4f} -> {rays / wall / 1e3:.1f} Mrays/s")


if __name__ == "__main__":
    main()
