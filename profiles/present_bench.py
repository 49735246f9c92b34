#!/usr/bin/env python3
"""What an embedder that SHOWS every frame pays (the reference blits its framebuffer every frame, src/gamelogic/world.rs:269-283):
C3's frame (1920x1080, depth 12, primary + shadow) delivered to host memory, frame after frame --
  host-target      vx_render with a host target (render, read back, wait: one after the other)
  present-rgba32f  vx_present_begin / vx_present_wait, one frame ahead: the read-back of frame k beside the kernel of frame k+1
  present-rgba8    the same with RGBA8 pixels (Framebuffer::as_image's format): a quarter of the bytes over PCIe

    python profiles/present_bench.py --format csvo
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--format", default="csvo")
    ap.add_argument("--frames", type=int, default=200)
    args = ap.parse_args()
    fmt = vra.SVO_ESVO if args.format == "esvo" else vra.SVO_CSVO
    W, H, depth = 1920, 1080, 12
    world = vra.World(fmt)
    st = world.build_heightfield(depth)
    svo = hip.Svo(fmt, world.size_in_bytes + (16 << 20))
    svo.set_materials(scenes.synthetic_materials())
    svo.set_textures(scenes.asset_textures(ROOT / "tests" / "golden" / "textures"), 6)
    svo.update(world)
    u = scenes.bench_camera(depth, st["h_max"], W, H, shadow_distance=3.0e38, render_shadows=True)
    out = {"workload": f"C3 frame to host memory every frame, {args.format.upper()}", "frames": args.frames}
    for _ in range(5):
        svo.render(u, W, H)
    t0 = time.perf_counter()
    for _ in range(args.frames // 4):
        svo.render(u, W, H)
    out["host_target_ms"] = round((time.perf_counter() - t0) * 1e3 / (args.frames // 4), 4)
    for name, f in (("present_rgba32f_ms", hip.VX_FORMAT_RGBA32F), ("present_rgba8_ms", hip.VX_FORMAT_RGBA8)):
        prev = svo.present_begin(u, W, H, f)
        for _ in range(8):
            nxt = svo.present_begin(u, W, H, f)
            svo.present_wait(prev, W, H, f)
            prev = nxt
        t0 = time.perf_counter()
        checksum = 0
        for _ in range(args.frames):
            nxt = svo.present_begin(u, W, H, f)
            img = svo.present_wait(prev, W, H, f)
            checksum += int(img[H // 2, W // 2, 0] > 0)  # touch the frame
            prev = nxt
        out[name] = round((time.perf_counter() - t0) * 1e3 / args.frames, 4)
        svo.present_wait(prev, W, H, f)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
