#!/usr/bin/env python3
"""What the walks inside voxels look like (CPU, oracle with its per-iteration frames): for the shadow rays of a low-resolution frame of the
depth-D CSVO terrain -- origins taken as the primary hit positions, which lie inside their voxels by construction -- how many iterations the
reference spends inside the voxel, how deep it gets below the voxel, whether it tests phantom leaves, crosses phantom chunk boundaries, and how
it ends. Statistics for the design of the lean excursion (round 4); not a parity tool.

    python profiles/round4/tools/excursion_stats.py --depth 13 --width 320 --height 180
"""
import argparse
import collections
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
from _pkg import load_package  # noqa: E402

vra = load_package()
from oracle import oracle as orc  # noqa: E402
from voxel_rs_amd import scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=13)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--height", type=int, default=180)
    args = ap.parse_args()
    world = vra.World(vra.SVO_CSVO)
    st = world.build_heightfield(args.depth)
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    scene = orc.OracleScene(vra.SVO_CSVO, world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    w, h = args.width, args.height
    u = scenes.bench_camera(args.depth, st["h_max"], w, h, shadow_distance=3.0e38)
    _, hits = scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), w, h)
    light = -np.asarray(u.light_dir[:], dtype=np.float32)
    n_rays = 0
    iters = collections.Counter()
    levels = collections.Counter()
    kinds = collections.Counter()
    first_child = collections.Counter()
    for y in range(h):
        for x in range(w):
            hit = hits[y, x]
            if not (hit["flags"] & 2):
                continue
            res, frames, n = scene.intersect(hit["pos"], light, -1.0, True, max_frames=1100)
            # the iteration in which the ray is led into the voxel: a leaf child that is reached with t_min == 0 (depth field == 1)
            k0 = None
            for k in range(len(frames)):
                f = frames[k]
                if f["is_leaf"] and f["is_child"] and f["t_min"] == 0.0 and f["parent_octant_idx"] == 1:
                    k0 = k
                    break
            if k0 is None:
                kinds["not_inside"] += 1
                continue
            n_rays += 1
            parent_scale = int(frames[k0]["scale"])
            k = k0 + 1
            inside = 1
            deepest = 0
            crossed = False
            leaf_tests = 0
            while k < len(frames) and frames[k]["scale"] < parent_scale:
                f = frames[k]
                inside += 1
                deepest = max(deepest, parent_scale - 1 - int(f["scale"]))
                crossed |= bool(f["crossed_boundary"]) and bool(f["is_child"])
                if f["is_leaf"] and f["is_child"] and f["t_min"] > 0.0:
                    leaf_tests += 1
                k += 1
            if k0 + 1 < len(frames):
                f1 = frames[k0 + 1]
                first_child["N0 has the origin's cell" if (f1["is_child"] and f1["scale"] < parent_scale) else "N0 lacks it / span empty"] += 1
            ended_inside = k >= len(frames)
            iters[inside] += 1
            levels[deepest] += 1
            kinds["crossed a phantom boundary"] += int(crossed)
            kinds["phantom leaf tests >= 1"] += int(leaf_tests > 0)
            kinds["ended inside (hit or miss)"] += int(ended_inside)
    print(f"depth {args.depth}, {w}x{h}: {n_rays} shadow rays led into their voxel; {kinds['not_inside']} not")
    print("iterations inside (incl. the PUSH):", sorted(iters.items()))
    print("mean:", sum(k * v for k, v in iters.items()) / max(1, n_rays))
    print("levels below the voxel's own node N0 reached:", sorted(levels.items()))
    print("first look at N0:", dict(first_child))
    for k, v in kinds.items():
        print(f"  {k}: {v} ({100.0 * v / max(1, n_rays):.1f} %)")


if __name__ == "__main__":
    main()
