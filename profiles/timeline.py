#!/usr/bin/env python3
"""Where a frame's time goes, wave by wave: when each persistent wave of one C3 frame started, found the sub-tile queue empty and left
(VX_TIMELINE=1; vx_timeline_read). One frame at a time.

    VX_TIMELINE=1 python profiles/timeline.py --format esvo [--hot 0|1]
"""
import argparse
import json
import os
import sys
from pathlib import Path

os.environ["VX_TIMELINE"] = "1"
ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("VX_LIB_DIR", str(ROOT / "voxel-rs_amd" / "lib" / "lib_tl"))  # the library's timeline build (make tl)
sys.path.insert(0, str(ROOT))
from _pkg import load_package  # noqa: E402

vra = load_package()
from voxel_rs_amd import hip, scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--format", default="esvo")
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--hot", default="1")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    args = ap.parse_args()
    os.environ["VX_HOT_FIRST"] = args.hot
    import numpy as np
    import torch

    fmt = vra.SVO_ESVO if args.format == "esvo" else vra.SVO_CSVO
    W, H = args.width, args.height
    world = vra.World(fmt)
    st = world.build_heightfield(args.depth)
    svo = hip.Svo(fmt, world.size_in_bytes + (16 << 20))
    svo.set_materials(scenes.synthetic_materials())
    svo.set_textures(scenes.asset_textures(ROOT / "tests" / "golden" / "textures"), 6)
    svo.update(world)
    svo.set_frames_in_flight(1)
    u = scenes.bench_camera(args.depth, st["h_max"], W, H, shadow_distance=3.0e38, render_shadows=True)
    image = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    svo.profile_enable(True)
    for _ in range(6):
        svo.render_device(u, W, H, image.data_ptr())
    svo.sync()
    ms, launches = svo.profile_read()
    svo.profile_enable(False)
    raw = svo.timeline()
    # row[6]: trips | ADVANCE-only trips << 20 | PUSH-only trips << 40 (the hand-scheduled loop's cheap tails)
    r6 = raw[:, 6].copy()
    adv_only, push_only = ((r6 >> np.uint64(20)) & np.uint64(0xfffff)).astype(np.float64), ((r6 >> np.uint64(40)) & np.uint64(0xfffff)).astype(np.float64)
    raw[:, 6] = r6 & np.uint64(0xfffff)
    t = raw.astype(np.float64)
    t0 = t[:, 0].min()
    start, empty, leave = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0  # microseconds
    q = lambda a: [round(float(np.percentile(a, p)), 1) for p in (0, 10, 50, 90, 99, 100)]
    r7 = t[:, 7].astype(np.uint64)
    wp, wt, wc = (r7 >> np.uint64(52)).astype(np.float64), ((r7 >> np.uint64(32)) & np.uint64(0xfffff)).astype(np.float64), (r7 & np.uint64(0xffffffff)).astype(np.float64)
    fit = [round(float(v), 1) for v in np.linalg.lstsq(np.stack([wp, wt], axis=1), wc, rcond=None)[0]] if wp.sum() > 0 else None
    print(json.dumps({"format": args.format, "hot_first": args.hot, "waves": int(len(t)), "percentiles": [0, 10, 50, 90, 99, 100],
                      "start_us": q(start), "queue_empty_us": q(empty[t[:, 1] > 0]), "exit_us": q(leave), "tail_us_per_wave": q((leave - empty)[t[:, 1] > 0]),
                      "subtiles_taken": q(t[:, 3].astype(np.uint64) & np.uint64(0xfffff)),
                      "service_phases_per_wave": q((t[:, 3].astype(np.uint64) >> np.uint64(20)) & np.uint64(0xfff)),
                      "us_in_service_phases_per_wave": q((t[:, 3].astype(np.uint64) >> np.uint64(32)).astype(np.float64) / 100.0), "kernel_us": round(float(leave.max()), 1),
                      "kernel_us_by_events_mean_of_launches": round(ms / max(launches, 1) * 1e3, 1),
                      "mean_wave_lifetime_us": round(float((leave - start).mean()), 1),
                      # shader-clock stamps: the clock the kernel ran at, and what a trip of the traversal loop costs the wave that makes it
                      "clock_mhz": q(t[:, 4] / np.maximum(t[:, 2] - t[:, 0], 1.0) * 100.0),
                      "loop_share_of_wave_life": q(t[:, 5] / np.maximum(t[:, 4], 1.0)),
                      "loop_trips_per_wave": q(t[:, 6]),
                      # of a frame's trips: every traversing lane ADVANCEs / every one PUSHes / both kinds (the merged tail, 22-31 instructions longer)
                      "trips_by_tail_advance_only_push_only_merged": [round(float(adv_only.sum() / max(t[:, 6].sum(), 1.0)), 3), round(float(push_only.sum() / max(t[:, 6].sum(), 1.0)), 3),
                                                                      round(float(1.0 - (adv_only.sum() + push_only.sum()) / max(t[:, 6].sum(), 1.0)), 3)],
                      "cycles_per_trip_as_a_wave_sees_it": q(t[:, 5] / np.maximum(t[:, 6], 1.0)),
                      "cycles_per_trip_mean": round(float(t[:, 5].sum() / max(t[:, 6].sum(), 1.0)), 1),
                      "simd_cycles_per_trip_at_4_waves": round(float(t[:, 5].sum() / max(t[:, 6].sum(), 1.0)) / 4.0, 1),
                      # the walks inside voxels (CSVO worlds): phases per wave, trips of the walk's loop (the slowest lane's iterations, summed over the
                      # phases), shader-clock cycles in the walks -- and the least-squares fit cycles = a * phases + b * trips over the waves
                      "walk_phases_per_wave": q(wp), "walk_trips_per_wave": q(wt), "walk_cycles_per_wave": q(wc), "walk_fit_cycles_per_phase_and_per_trip": fit}))


if __name__ == "__main__":
    main()
