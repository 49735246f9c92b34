#!/usr/bin/env python3
"""What handing a batch's stragglers over would buy (CPU, oracle): the C3 frame's per-ray iteration counts; a lockstep wave runs a sub-tile's 64
primary rays until the last has ended, then its shadow rays. With a threshold T the wave leaves the loop as soon as at most T lanes still
traverse: those rays are written down (pixel / origin), their lanes freed, and run again from the start later, 64 of a wave's stragglers at a time
(no second hand-over). Trips of the traversal loop = the sum over batches of the trip at which the batch ends.

    python profiles/round4/tools/straggler_sim.py [--width 960 --height 544]
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
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--height", type=int, default=544)
    ap.add_argument("--waves", type=int, default=1024, help="waves that share the frame (a wave's stragglers wait until it has 64 of them)")
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
    total = hits["steps"].astype(np.int64)
    shadow = np.where((hits["flags"] & 2) != 0, total - prim, 0)
    # sub-tiles in screen order, dealt round-robin to the waves
    tiles = [(by, bx) for by in range(0, H - 7, 8) for bx in range(0, W - 7, 8)]
    useful = int(prim.sum() + shadow.sum())
    for T in (0, 1, 2, 4, 8, 12, 16):
        trips = 0
        deferred = 0
        pools = [[] for _ in range(args.waves)]
        for k, (by, bx) in enumerate(tiles):
            pool = pools[k % args.waves]
            p = prim[by:by + 8, bx:bx + 8].ravel()
            s = shadow[by:by + 8, bx:bx + 8].ravel()
            order = np.sort(p)[::-1]
            cut = int(order[T]) if T < 64 else 0
            trips += cut
            keep = p <= cut
            # a handed-over primary ray is run again whole, and its shadow ray with it (one after the other in the later batch)
            for i in np.nonzero(~keep)[0]:
                pool.append(int(p[i]))
                if s[i] > 0:
                    pool.append(int(s[i]))
            deferred += int((~keep).sum())
            sk = s[keep & (s > 0)]
            if len(sk):
                order = np.sort(sk)[::-1]
                cut = int(order[T]) if T < len(order) else 0
                trips += cut
                for v in sk[sk > cut]:
                    pool.append(int(v))
                deferred += int((sk > cut).sum())
        for pool in pools:
            for i in range(0, len(pool), 64):
                trips += max(pool[i:i + 64])
        print(f"T = {T:2d}: {trips} trips ({trips / max(1, base if T else trips):.3f} of lockstep), {deferred} rays handed over ({100.0 * deferred / (2 * len(tiles) * 64):.1f} % of the lane slots), "
              f"lane slots used {useful / (trips * 64):.3f}") if T else None
        if T == 0:
            base = trips
            print(f"{W}x{H}: {len(tiles)} sub-tiles, lockstep: {trips} trips, lane slots used {useful / (trips * 64):.3f}")


if __name__ == "__main__":
    main()
