#!/usr/bin/env python3
"""How much of a pass-sorted frame's saving survives a moving camera (CPU, oracle): per-pixel iteration counts of two frames of the C3 scene a
small turn apart; the 16x16 blocks of the second frame cut into four passes of 64 (a) by the frame's own costs (what a still view knows), (b) by
the first frame's costs at the same pixels (round 3 under motion), (c) by the first frame's costs where the second frame's camera sees them (the
rotation reprojected, nearest pixel). Trips of a lockstep wave = the sum over passes of the pass's dearest ray.

    python profiles/round4/tools/reprojected_cost_sim.py --degrees 0.5
"""
import argparse
import math
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
    ap.add_argument("--degrees", type=float, default=0.5)
    args = ap.parse_args()
    W, H, depth = args.width, args.height, args.depth
    world = vra.World(vra.SVO_CSVO)
    st = world.build_heightfield(depth)
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    scene = orc.OracleScene(vra.SVO_CSVO, world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    n = float(1 << depth)

    def uniforms(deg):
        a = math.radians(deg)
        fwd = (0.6 * math.cos(a) - 0.7 * math.sin(a), -0.35, 0.6 * math.sin(a) + 0.7 * math.cos(a))
        return scenes.render_params_to_uniforms((0.5 * n, st["h_max"] + 0.05 * n, 0.5 * n), fwd, (0.0, 1.0, 0.0), math.radians(72.0), W / H, 0.3, (-1.0, -1.0, -1.0), True, 3.0e38)

    u0, u1 = uniforms(0.0), uniforms(args.degrees)
    c0 = scene.render(orc.Uniforms.from_buffer_copy(bytes(u0)), W, H)[1]["steps"].astype(np.int64)
    c1 = scene.render(orc.Uniforms.from_buffer_copy(bytes(u1)), W, H)[1]["steps"].astype(np.int64)
    # reprojection: pixel of frame 1 -> the pixel of frame 0 that looked in its direction
    tan_half = math.tan(math.radians(72.0) / 2)
    aspect = W / H
    R0 = np.array(u0.view[:], dtype=np.float64).reshape(4, 4).T[:3, :3]  # camera-to-world (column-major in the uniforms)
    R1 = np.array(u1.view[:], dtype=np.float64).reshape(4, 4).T[:3, :3]
    ys, xs = np.mgrid[0:H, 0:W]
    cx = (xs / W * 2 - 1) * aspect * tan_half
    cy = (ys / H * 2 - 1) * tan_half
    d1 = np.stack([cx, cy, -np.ones_like(cx)], axis=-1) @ R1.T  # world directions of frame 1's pixels
    d0 = d1 @ R0  # in frame 0's camera
    ok = d0[..., 2] < 0
    px = np.floor((d0[..., 0] / -d0[..., 2] / (aspect * tan_half) + 1) * 0.5 * W + 0.5).astype(np.int64)
    py = np.floor((d0[..., 1] / -d0[..., 2] / tan_half + 1) * 0.5 * H + 0.5).astype(np.int64)
    ok &= (px >= 0) & (px < W) & (py >= 0) & (py < H)
    reproj = np.where(ok, c0[np.clip(py, 0, H - 1), np.clip(px, 0, W - 1)], 0)

    def trips(cost, key):
        total = 0
        for by in range(0, H - 15, 16):
            for bx in range(0, W - 15, 16):
                c = cost[by:by + 16, bx:bx + 16].ravel()
                k = key[by:by + 16, bx:bx + 16].ravel()
                order = np.argsort(k, kind="stable")
                total += sum(int(c[order[i:i + 64]].max()) for i in range(0, 256, 64))
        return total

    def trips_subtiles(cost):
        total = 0
        for by in range(0, H - 15, 16):
            for bx in range(0, W - 15, 16):
                for sy in (0, 8):
                    for sx in (0, 8):
                        total += int(cost[by + sy:by + sy + 8, bx + sx:bx + sx + 8].max())
        return total

    base = trips_subtiles(c1)
    print(f"turn {args.degrees} degrees, {W}x{H}: a pixel is {math.degrees(2 * tan_half / H):.3f} degrees")
    print("  8x8 sub-tiles (unsorted)            :", base, "trips, 1.000")
    for name, key in (("own costs (a still view)", c1), ("the other frame's, same pixels", c0), ("the other frame's, reprojected", reproj)):
        t = trips(c1, key)
        print(f"  passes by {name:33s}: {t} trips, {t / base:.3f}")
    corr = np.corrcoef(c1[ok].ravel(), reproj[ok].ravel())[0, 1]
    print(f"  correlation of the frame's costs with the reprojected ones: {corr:.3f}; with the same pixels' of the other frame: {np.corrcoef(c1.ravel(), c0.ravel())[0, 1]:.3f}")


if __name__ == "__main__":
    main()
