"""Worker of tests/test_hip_parity.py: one process that loads ONE build of the library (VX_LIB_DIR: the measurement build, lib/lib_tl, honours the knobs
of experiments -- VX_REFILL_MIN, VX_HOT_FIRST, VX_QUEUE_STRIPE, VX_TILE_STRIP, VX_WAVES_PER_CU --, the product build does not), renders a fixed set of
frames of a small world under the environment it was started with, and prints their digests as one JSON line. A process loads one of the two builds,
and the knobs are read when a context is created: hence a process per setting.

    VX_LIB_DIR=voxel-rs_amd/lib/lib_tl VX_REFILL_MIN=7 python tests/knob_worker.py csvo
"""
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    fmt_name = sys.argv[1]
    import torch

    from _pkg import load_package

    vra = load_package()
    from voxel_rs_amd import hip, scenes

    fmt = vra.SVO_ESVO if fmt_name == "esvo" else vra.SVO_CSVO
    world = vra.World(fmt)
    st = world.build_heightfield(8, threads=4)
    w, h = 333, 227  # (an odd size: a last strip narrower than the others, edge tiles)
    u = scenes.bench_camera(8, st["h_max"], w, h, shadow_distance=3.0e38)
    svo = hip.Svo(fmt, world.size_in_bytes + (1 << 20))
    svo.set_materials(scenes.synthetic_materials())
    svo.set_textures(scenes.synthetic_textures(), 6)
    svo.update_full(world)
    sha = lambda b: hashlib.sha256(b).hexdigest()
    img, hits = svo.render(u, w, h, want_hits=True)
    out = {"hits": sha(img.tobytes() + hits.tobytes())}
    for k in range(7):  # the context's own stream: from the third frame on through the cost-ordered table
        out["own %d" % k] = sha(svo.render(u, w, h)[0].tobytes())
    t = [torch.zeros((h, w, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for k in range(4):
        svo.render_device(u, w, h, t[k & 1].data_ptr())
    svo.sync()
    out["in flight"] = sha(t[0].cpu().numpy().tobytes() + t[1].cpu().numpy().tobytes())
    n = hip.local_tile_count(w, h, 1, 3)
    lst = torch.zeros((n * 1024, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    svo.render_device(u, w, h, lst.data_ptr(), tile_rank=1, tile_count=3)
    svo.sync()
    out["rank 1 of 3"] = sha(lst.cpu().numpy().tobytes())
    # (a frame of several sub-tiles per resident wave: only then do the refill and service thresholds show in the scheduling counters)
    cw, ch = 1280, 720
    counters = svo.render_counters(scenes.bench_camera(8, st["h_max"], cw, ch, shadow_distance=3.0e38), cw, ch)
    occupancy = ("wave_steps", "services", "refills", "tail_wave_steps", "tail_iterations")
    out["counters"] = {k: v for k, v in counters.items() if k not in occupancy}
    # how the instrumented kernel scheduled its lanes (not what the rays did): moves with the refill / service thresholds wherever they are honoured
    out["occupancy"] = {k: counters[k] for k in occupancy}
    out["knobs"] = svo.knobs()  # (what the context runs with: the measurement build reads the experiments' environment variables, the product build does not)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
