#!/usr/bin/env python3
"""What ONE early exit per batch would buy (CPU, oracle): the C3 frame's per-ray iteration counts. Lockstep: a wave runs a batch (a sub-tile's 64 primary
rays, then its shadow rays) until the last ray has ended. Rule K: the wave leaves its loop when every ray carried over from the batch before has ended and
at most K of the batch's own rays still traverse; those K are carried into the next batch (each ray is carried once at most, a service phase per batch as
in lockstep). Trips of the loop per policy, and the lane slots they use. Lanes are not rationed here (a batch is its 64 rays whatever is carried): an
upper bound of the gain.

    python profiles/round5/tools/carry_stragglers_sim.py [--width 1920 --height 1080]
"""
import argparse
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
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--waves", type=int, default=4096)
    args = ap.parse_args()
    W, H, depth = args.width, args.height, args.depth
    world = vra.World(vra.SVO_CSVO)
    st = world.build_heightfield(depth)
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    scene = orc.OracleScene(vra.SVO_CSVO, world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    u = scenes.bench_camera(depth, st["h_max"], W, H, shadow_distance=3.0e38)
    hits = scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), W, H)[1]
    u0 = scenes.bench_camera(depth, st["h_max"], W, H, shadow_distance=3.0e38, render_shadows=False)
    prim = scene.render(orc.Uniforms.from_buffer_copy(bytes(u0)), W, H)[1]["steps"].astype(np.int64)
    shadow = np.where((hits["flags"] & 2) != 0, hits["steps"].astype(np.int64) - prim, 0)
    tiles = [(by, bx) for by in range(0, H - 7, 8) for bx in range(0, W - 7, 8)]
    useful = int(prim.sum() + shadow.sum())
    # a wave's sub-tiles: the queue's order is not the screen's, and which wave gets which sub-tile depends on timing; round-robin is close enough
    per_wave = [tiles[w::args.waves] for w in range(args.waves)]
    base = None
    for K in (0, 2, 4, 8, 12, 16, 24):
        trips = 0
        batches = 0
        for mine in per_wave:
            old = np.zeros(0, dtype=np.int64)
            for by, bx in mine:
                for rays in (prim[by:by + 8, bx:bx + 8].ravel(), shadow[by:by + 8, bx:bx + 8].ravel()):
                    rays = rays[rays > 0]
                    if len(rays) == 0 and len(old) == 0:
                        continue
                    own = np.sort(rays)[::-1]
                    leave = int(own[K]) if K < len(own) else 0   # at most K of the batch's own rays still traverse
                    if len(old):
                        leave = max(leave, int(old.max()))       # ... and nothing older does
                    trips += leave
                    batches += 1
                    old = own[own > leave] - leave
            if len(old):
                trips += int(old.max())
        if K == 0:
            base = trips
        print(f"K = {K:2d}: {trips} trips ({trips / base:.3f} of lockstep), {batches} batches, lane slots used {useful / (trips * 64):.3f}")


if __name__ == "__main__":
    main()
