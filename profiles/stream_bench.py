#!/usr/bin/env python3
"""C4-style streaming measurement (NOT the headline bench): a camera flies over the synthetic
terrain while the chunk loader streams chunks in (nearest first, at most 400 events per commit as worldsvo.rs:139),
one commit and one frame per step.

    python profiles/stream_bench.py --format csvo --scene-depth 12 --radius 40 --width 3840 --height 2160 --frames 80

Prints one JSON line: Mrays/s while streaming and after the stream has drained, bytes / ranges per commit, host time per
commit split into chunk build (worker threads), apply (set_leaf + root) and commit (staging write + vx_commit), H2D rate."""
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--format", default="csvo")
    ap.add_argument("--scene-depth", type=int, default=12)
    ap.add_argument("--radius", type=int, default=40)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=80)
    ap.add_argument("--speed", type=float, default=2.0, help="blocks per frame along +x")
    ap.add_argument("--events", type=int, default=400)
    ap.add_argument("--capacity-mb", type=int, default=2000)
    ap.add_argument("--pipelined", type=int, default=1, help="1 = VX_COMMIT_PIPELINED during the flight (the commit worker), 0 = inline commits")
    return ap.parse_args(argv)


def run(args):
    """The measurement, as a dict (bench.py calls this for its `configs.C4_streamed` object; `args` = parse_args([...]))."""
    import torch

    fmt = vra.SVO_ESVO if args.format == "esvo" else vra.SVO_CSVO
    n = float(1 << args.scene_depth)
    y_chunks = max(2, int(n / 4 / 32) + 1)  # the terrain is at most 2^depth / 4 high
    s = host.WorldStreamer(fmt, args.scene_depth, args.radius, 0, y_chunks)
    svo = hip.Svo(fmt, args.capacity_mb * 1000 * 1000)
    svo.set_materials(scenes.synthetic_materials())
    svo.set_textures(scenes.synthetic_textures(), 6)
    W, H = args.width, args.height
    image = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    def ground(x, z):
        return float(host.lib().vxh_scene_height(args.scene_depth, 0x5EED0001, int(x), int(z)))

    # fly 90 blocks above the highest ground along the path, so that the terrain is inside the loader's vertical radius
    path_x = [0.3 * n + args.speed * i for i in range(args.frames + 1)]
    eye = [0.3 * n, max(ground(x, 0.5 * n) for x in path_x) + 90.0, 0.5 * n]
    # phase 1: initial fill around the start position (large commits, no frames in between)
    t0 = time.perf_counter()
    s.move_to(*eye)
    fill = dict(events=0, bytes=0, commits=0, build_us=0, apply_us=0, commit_us=0)
    while True:
        st = s.pump(svo._h, args.events, wait=False)  # what the background workers have finished, at most --events per commit
        for k in ("events", "bytes", "build_us", "apply_us", "commit_us"):
            fill[k] += st[k]
        fill["commits"] += 1 if st["events"] else 0
        if st["pending"] == 0:
            break
        if st["events"] == 0:
            time.sleep(0.0002)
    svo.sync()
    fill_s = time.perf_counter() - t0
    # phase 2: fly along +x, one commit of <= args.events events and one frame per step
    rows = []
    svo.profile_enable(True)
    svo.set_commit_mode(bool(args.pipelined))
    for f in range(args.frames):
        eye[0] += args.speed
        t0 = time.perf_counter()
        s.move_to(*eye)
        st = s.pump(svo._h, args.events, wait=False)
        host_ms = (time.perf_counter() - t0) * 1e3
        cam = s.to_svo(eye)
        u = scenes.render_params_to_uniforms(cam, (0.6, -0.35, 0.7), (0.0, 1.0, 0.0), math.radians(72.0), W / H, 0.3, (-1.0, -1.0, -1.0), True, 3.0e38)
        svo.render_device(u, W, H, image.data_ptr())
        step_ms = (time.perf_counter() - t0) * 1e3  # everything the frame loop's thread did this step
        ms, launches = svo.profile_read()
        rows.append(dict(st, host_ms=host_ms, step_ms=step_ms, kernel_ms=ms / max(launches, 1), frame=f))
    # phase 3: stand still until the queue has drained, then a few settled frames
    settled = []
    svo.set_commit_mode(False)
    while True:
        st = s.pump(svo._h, args.events)
        if st["pending"] == 0:
            break
    for _ in range(10):
        svo.render_device(u, W, H, image.data_ptr())
        ms, launches = svo.profile_read()
        settled.append(dict(kernel_ms=ms / max(launches, 1)))
    svo.profile_enable(False)
    last = svo.render_counters(u, W, H)
    rays = last["rays"]
    streaming = [r for r in rows if r["events"] > 0]
    commits = [r for r in streaming if r["bytes"] > 0]
    out = {
        "workload": f"{W}x{H} primary+shadow, depth-{args.scene_depth} terrain streamed by the chunk loader (radius {args.radius} chunks, <= {args.events} events per commit), {args.format.upper()}",
        "initial_fill": {"events": fill["events"], "MB": round(fill["bytes"] / 1e6, 1), "seconds": round(fill_s, 2), "commits": fill["commits"],
                         "build_s": round(fill["build_us"] / 1e6, 2), "apply_s": round(fill["apply_us"] / 1e6, 2), "commit_s": round(fill["commit_us"] / 1e6, 3),
                         "commit_GBps": round(fill["bytes"] / max(fill["commit_us"], 1) / 1e3, 2)},
        "frames": len(rows), "streaming_frames": len(streaming), "pending_after_flight": rows[-1]["pending"], "resident_chunks": s.resident_chunks, "arena_MB": round(rows[-1]["arena_bytes"] / 1e6, 1),
        "rays_last_frame": int(rays), "iterations_last_frame": int(last["iterations"]),
        "kernel_ms_streaming_median": round(statistics.median(r["kernel_ms"] for r in streaming), 3) if streaming else None,
        "kernel_ms_settled_median": round(statistics.median(r["kernel_ms"] for r in settled), 3),
        "Mrays_per_s_settled": round(rays / statistics.median(r["kernel_ms"] for r in settled) / 1e3, 1),
        "commit_MB_median": round(statistics.median(r["bytes"] for r in commits) / 1e6, 2) if commits else None,
        "commit_ranges_median": int(statistics.median(r["ranges"] for r in commits)) if commits else None,
        # move_to (chunk loader) + pump (apply finished chunks, root, staging write, vx_commit); chunks are built by background workers
        "host_ms_per_step_median": round(statistics.median(r["host_ms"] for r in streaming), 3) if streaming else None,
        "commit_mode": "pipelined" if args.pipelined else "inline",
        "host_ms_per_step_with_render_call_median": round(statistics.median(r["step_ms"] for r in streaming), 3) if streaming else None,
        "host_ms_per_step_max": round(max(r["host_ms"] for r in streaming), 3) if streaming else None,
        # the three dearest steps of the flight: what the frame loop's thread spent where (pump's own clocks), and how many events the step applied
        "host_ms_dearest_steps": [dict(frame=r["frame"], host_ms=round(r["host_ms"], 2), apply_ms=round(r["apply_us"] / 1e3, 2), commit_ms=round(r["commit_us"] / 1e3, 2),
                                       events=r["events"], MB=round(r["bytes"] / 1e6, 2)) for r in sorted(streaming, key=lambda r: -r["host_ms"])[:3]],
        "build_ms_median": round(statistics.median(r["build_us"] for r in commits) / 1e3, 2) if commits else None,
        "apply_ms_median": round(statistics.median(r["apply_us"] for r in commits) / 1e3, 2) if commits else None,
        "commit_ms_median": round(statistics.median(r["commit_us"] for r in commits) / 1e3, 2) if commits else None,
        "commit_GBps_median": round(statistics.median(r["bytes"] / max(r["commit_us"], 1) / 1e3 for r in commits), 2) if commits else None,
    }
    svo.close()
    return out


if __name__ == "__main__":
    print(json.dumps(run(parse_args())))
