#!/usr/bin/env python3
"""The first commit of a large world, timed: the depth-14 terrain, vx_commit of everything (image build + upload), with the library's own split on stderr
(VX_COMMIT_TIMING=1).   python profiles/round6/first_commit.py --format csvo [--depth 14] [--repeats 2]"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

os.environ["VX_COMMIT_TIMING"] = "1"
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from _pkg import load_package  # noqa: E402

vra = load_package()
from voxel_rs_amd import hip, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--format", default="csvo")
    ap.add_argument("--depth", type=int, default=14)
    ap.add_argument("--repeats", type=int, default=2)
    args = ap.parse_args()
    fmt = vra.SVO_ESVO if args.format == "esvo" else vra.SVO_CSVO
    t0 = time.perf_counter()
    world = vra.World(fmt)
    st = world.build_heightfield(args.depth)
    build_s = time.perf_counter() - t0
    for _ in range(args.repeats):
        svo = hip.Svo(fmt, world.size_in_bytes + (16 << 20))
        svo.set_materials(scenes.synthetic_materials())
        svo.set_textures(scenes.synthetic_textures(), 6)
        t0 = time.perf_counter()
        svo.update_full(world)
        svo.sync()
        commit_s = time.perf_counter() - t0
        print(json.dumps({"format": args.format, "depth": args.depth, "world_MB": round(world.size_in_bytes / 1e6, 1), "scene_build_s": round(build_s, 2),
                          "first_commit_s": round(commit_s, 3), "image": svo.image_info()}), flush=True)
        svo.close()


if __name__ == "__main__":
    main()
