#!/usr/bin/env python3
"""One frame at a time: which waves of a C3 frame start late, and what that costs (the measurement build's wave timeline).
    VX_TIMELINE=1 python profiles/round5/late_waves.py --format csvo"""
import argparse, json, os, sys
from pathlib import Path
os.environ["VX_TIMELINE"] = "1"
ROOT = Path(__file__).resolve().parents[2]
os.environ.setdefault("VX_LIB_DIR", str(ROOT / "voxel-rs_amd" / "lib" / "lib_tl"))
sys.path.insert(0, str(ROOT))
from _pkg import load_package  # noqa: E402
vra = load_package()
from voxel_rs_amd import hip, scenes  # noqa: E402

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--format", default="csvo"); args = ap.parse_args()
    import numpy as np, torch
    fmt = vra.SVO_ESVO if args.format == "esvo" else vra.SVO_CSVO
    W, H = 1920, 1080
    world = vra.World(fmt); st = world.build_heightfield(12)
    svo = hip.Svo(fmt, world.size_in_bytes + (16 << 20))
    svo.set_materials(scenes.synthetic_materials()); svo.set_textures(scenes.asset_textures(ROOT / "tests" / "golden" / "textures"), 6)
    svo.update(world); svo.set_frames_in_flight(1)
    u = scenes.bench_camera(12, st["h_max"], W, H, shadow_distance=3.0e38, render_shadows=True)
    image = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda"); torch.cuda.synchronize()
    rows = []
    for frame in range(8):
        svo.render_device(u, W, H, image.data_ptr()); svo.sync()
        t = svo.timeline().astype(np.float64)
        t0 = t[:, 0].min()
        start, leave = (t[:, 0] - t0) / 100.0, (t[:, 2] - t0) / 100.0
        late = np.nonzero(start > 100.0)[0]
        rows.append({"frame": frame, "waves": int(len(t)), "kernel_us": round(float(leave.max()), 1), "started_after_30us": int((start > 30).sum()), "started_after_100us": int(len(late)),
                     "late_wave_ids": late[:24].tolist(), "late_starts_us": [round(float(x), 1) for x in start[late][:24]], "late_exits_us": [round(float(x), 1) for x in leave[late][:24]],
                     "last_exit_among_prompt_waves_us": round(float(leave[start <= 100.0].max()), 1)})
    for r in rows: print(json.dumps(r))

if __name__ == "__main__":
    main()
