#!/usr/bin/env python3
"""Lockstep trips of a frame by the SHAPE of a wave's 64 pixels (CPU, oracle): a sub-tile's rays are traversed until its longest primary ray
has ended, then until its longest shadow ray has -- trips = the sum over sub-tiles of those two maxima. 8x8 (what the kernel uses) against
16x4, 32x2, 64x1 and the upright shapes, for the bench's C3 view.

    python profiles/round4/tools/subtile_shape_sim.py [--width 1920 --height 1080]
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
    args = ap.parse_args()
    W, H, depth = args.width, args.height, args.depth
    world = vra.World(vra.SVO_CSVO)
    st = world.build_heightfield(depth)
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    scene = orc.OracleScene(vra.SVO_CSVO, world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    u = scenes.bench_camera(depth, st["h_max"], W, H, shadow_distance=3.0e38, render_shadows=True)
    both = scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), W, H)[1]["steps"].astype(np.int64)
    u0 = scenes.bench_camera(depth, st["h_max"], W, H, shadow_distance=3.0e38, render_shadows=False)
    prim = scene.render(orc.Uniforms.from_buffer_copy(bytes(u0)), W, H)[1]["steps"].astype(np.int64)
    shad = both - prim
    Hc, Wc = H - H % 64, W - W % 64
    prim, shad = prim[:Hc, :Wc], shad[:Hc, :Wc]
    useful = int(prim.sum() + shad.sum())
    print(f"{Wc}x{Hc}: iterations {useful} ({prim.mean():.1f} primary + {shad.mean():.1f} shadow a pixel)")
    for w, h in ((8, 8), (16, 4), (32, 2), (64, 1), (4, 16), (2, 32), (1, 64)):
        p = prim.reshape(Hc // h, h, Wc // w, w).max(axis=(1, 3))
        s = shad.reshape(Hc // h, h, Wc // w, w).max(axis=(1, 3))
        trips = int(p.sum() + s.sum())
        print(f"  {w:2d} x {h:2d}: trips {trips}  lane slots used {useful / (64.0 * trips):.3f}  (8x8 = 1: {trips / 1.0:.0f})")


if __name__ == "__main__":
    main()
