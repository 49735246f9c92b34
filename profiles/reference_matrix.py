#!/usr/bin/env python3
"""The reference harness's matrix (benchmark-ingame.py:86-93): render distance {10, 20, 30, 40} x shadows on / off x LOD on / off x {esvo, csvo}, fov 80, at
1920x1080 -- on THIS build's streamed synthetic terrain (the reference flies over a Minecraft world with Perlin terrain that cannot be reproduced here: un-vendored
`noise`, no world file), a still camera once "all chunks loaded" (benchmark-ingame.py:27-37), frames presented one after the other for --seconds (the reference
samples 20 s). Columns are the reference's own (src/gamelogic/benchmark.rs:196-207): fps avg / med, frame_time_ms avg / med, svo_size_mb, plus the time the
world took to stream in (its `serialize_world` traces have no counterpart: chunks are serialized by the streamer's workers). A builder-run table for DESIGN.md,
not the driver's bench line; the reference published no numbers to put beside it.

    python profiles/reference_matrix.py [--seconds 3] [--distances 10 20 30 40] > profiles/round6/reference_matrix.jsonl
"""
import argparse
import json
import math
import statistics
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from _pkg import load_package  # noqa: E402

vra = load_package()
from voxel_rs_amd import hip, host, scenes  # noqa: E402


def one_case(args, torch, fmt_name, distance, shadows, no_lod):
    fmt = vra.SVO_ESVO if fmt_name == "esvo" else vra.SVO_CSVO
    depth = args.scene_depth
    n = float(1 << depth)
    y_chunks = max(2, int(n / 4 / 32) + 1)
    s = host.WorldStreamer(fmt, depth, distance, 0, y_chunks, no_lod=no_lod)
    svo = hip.Svo(fmt, args.gpu_buffer_mb * 1000 * 1000)  # (--gpu-buffer-size=3000 in the reference's command line)
    svo.set_materials(scenes.synthetic_materials())
    svo.set_textures(scenes.asset_textures(ROOT / "tests" / "golden" / "textures"), 6)
    W, H = args.width, args.height
    images = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    ground = float(host.lib().vxh_scene_height(depth, 0x5EED0001, int(0.5 * n), int(0.5 * n)))
    eye = [0.5 * n, ground + 24.0, 0.5 * n]  # (the reference stands on a hill: --pos -644 97 120)
    t0 = time.perf_counter()
    s.move_to(*eye)
    while True:
        st = s.pump(svo._h, 400, wait=False)  # (<= 400 events a commit: worldsvo.rs:139)
        if st["pending"] == 0:
            break
        if st["events"] == 0:
            time.sleep(0.0002)
    svo.sync()
    load_s = time.perf_counter() - t0  # "all chunks loaded"
    cam = s.to_svo(eye)
    # --rot -1 165 0: level, looking about south; fov 80
    yaw = math.radians(165.0)
    fwd = (math.sin(yaw), -0.0175, -math.cos(yaw))
    u = scenes.render_params_to_uniforms(cam, fwd, (0.0, 1.0, 0.0), math.radians(80.0), W / H, 0.3, (-1.0, -1.0, -1.0), shadows, 500.0)  # (shadow distance: world.rs:105-108)
    for i in range(20):
        svo.render_device(u, W, H, images[i % 2].data_ptr())
    svo.sync()
    # the reference's loop presents every frame (svo.rs:220-228 waits for the fence): a frame is issued when the one before it has been waited for
    frame_s = []
    t_end = time.perf_counter() + args.seconds
    i = 0
    while time.perf_counter() < t_end:
        t0 = time.perf_counter()
        svo.render_device(u, W, H, images[i % 2].data_ptr())
        svo.sync()
        frame_s.append(time.perf_counter() - t0)
        i += 1
    # ... and what the device does with two frames in flight (this library's default)
    t0 = time.perf_counter()
    k = 200
    for j in range(k):
        svo.render_device(u, W, H, images[j % 2].data_ptr())
    svo.sync()
    in_flight_ms = (time.perf_counter() - t0) / k * 1e3
    c = svo.render_counters(u, W, H)
    stats = svo.get_stats()
    out = {"svo_type": fmt_name, "render_distance": distance, "render_shadows": shadows, "no_lod": no_lod,
           "fps": {"avg": round(len(frame_s) / sum(frame_s), 1), "med": round(1.0 / statistics.median(frame_s), 1)},
           "frame_time_ms": {"avg": round(sum(frame_s) / len(frame_s) * 1e3, 4), "med": round(statistics.median(frame_s) * 1e3, 4)},
           "svo_size_mb": round(stats["used_bytes"] / 1024 / 1024, 2), "all_chunks_loaded_s": round(load_s, 2), "resident_chunks": s.resident_chunks,
           "frame_time_ms_two_in_flight": round(in_flight_ms, 4), "rays_per_frame": int(c["rays"]), "Mrays_per_s_two_in_flight": round(c["rays"] / in_flight_ms / 1e3, 1),
           "image": svo.image_info()}
    svo.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--distances", type=int, nargs="+", default=[10, 20, 30, 40])
    ap.add_argument("--formats", nargs="+", default=["esvo", "csvo"])
    ap.add_argument("--scene-depth", type=int, default=14)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--gpu-buffer-mb", type=int, default=3000)
    args = ap.parse_args()
    import torch

    for fmt_name in args.formats:
        for no_lod in (True, False):
            for shadows in (True, False):
                for distance in args.distances:
                    print(json.dumps(one_case(args, torch, fmt_name, distance, shadows, no_lod)), flush=True)


if __name__ == "__main__":
    main()
