#!/usr/bin/env python3
"""Renders one frame of a heightfield world three times -- from the traversal image, from the world's own bytes and from the
image layout for more than 4 GiB (VX_WIDE_IMAGE=1) -- and compares the images bit for bit (hit records too). A size-independent parity property: at depth 13 the CSVO image is 2.2 GB, i.e.
its pointers use the upper half of the 32-bit offset range.

    python profiles/image_vs_bytes.py --format csvo --depth 13
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from _pkg import load_package  # noqa: E402

vra = load_package()
from voxel_rs_amd import hip, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--format", default="csvo")
    ap.add_argument("--depth", type=int, default=13)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--wide", default="1", help="VX_WIDE_IMAGE for the third render: 1 = octant-index layout, 2 = and placed beyond 4 GiB")
    args = ap.parse_args()
    import numpy as np

    fmt = vra.SVO_ESVO if args.format == "esvo" else vra.SVO_CSVO
    world = vra.World(fmt)
    st = world.build_heightfield(args.depth)
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    u = scenes.bench_camera(args.depth, st["h_max"], args.width, args.height, shadow_distance=3.0e38)
    out = {"format": args.format, "depth": args.depth, "world_bytes": world.size_in_bytes, "VX_WIDE_IMAGE": args.wide}
    frames = {}
    for name, env, wide in (("image", "1", "0"), ("bytes", "0", "0"), ("wide", "1", args.wide)):
        os.environ["VX_TRAVERSAL_IMAGE"] = env
        os.environ["VX_WIDE_IMAGE"] = wide
        svo = hip.Svo(fmt, world.size_in_bytes + (16 << 20))
        svo.set_materials(mats)
        svo.set_textures(tex, 6)
        t0 = time.time()
        svo.update_full(world)
        out[f"commit_s_{name}"] = round(time.time() - t0, 3)
        frames[name] = svo.render(u, args.width, args.height, want_hits=True)
        del svo
    (ia, ha), (ib, hb), (iw, hw) = frames["image"], frames["bytes"], frames["wide"]
    out["images_identical"] = bool(np.array_equal(ia, ib) and np.array_equal(ia, iw))
    out["hit_records_identical"] = bool(ha.tobytes() == hb.tobytes() and ha.tobytes() == hw.tobytes())
    out["hits"] = int((ha["flags"] & 1).sum())
    print(json.dumps(out))
    if not (out["images_identical"] and out["hit_records_identical"]):
        raise SystemExit(1)


if __name__ == "__main__":
    main()
