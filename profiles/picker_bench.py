#!/usr/bin/env python3
"""C1 of BASELINE.json: 256x256 picker rays against one 32^3 chunk (plus the same rays against the depth-12 bench scene),
through vx_raycast (host arrays in, host arrays out, synchronous like the reference's fence wait) and through the CPU oracle."""
import ctypes as C
import json
import math
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from _pkg import load_package  # noqa: E402

vra = load_package()
from oracle import oracle as orc  # noqa: E402  (CPU baseline leg only)
from voxel_rs_amd import hip, host, scenes  # noqa: E402


def tasks_for(eye, target, n, max_dst):
    u = scenes.render_params_to_uniforms(tuple(eye), tuple(np.float32(target) - np.float32(eye)), (0.0, 1.0, 0.0), math.radians(72.0), 1.0, 0.3,
                                         (-1.0, -1.0, -1.0), False, 500.0)
    ou = orc.Uniforms.from_buffer_copy(bytes(u))
    tasks = np.zeros(n * n, dtype=orc.PICKER_TASK_DTYPE)
    ro, rd = (C.c_float * 3)(), (C.c_float * 3)()
    for y in range(n):
        for x in range(n):
            orc.lib().or_primary_ray(C.byref(ou), n, n, x, y, C.byref(ro), C.byref(rd))
            t = tasks[y * n + x]
            t["max_dst"], t["pos"], t["dir"] = max_dst, list(ro), list(rd)
    return tasks


def measure(fmt, world, tasks, label):
    tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
    svo = hip.Svo(fmt, world.size_in_bytes + (4 << 20))
    svo.set_materials(mats)
    svo.set_textures(tex, 6)
    svo.update(world)
    got = svo.raycast(tasks)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        svo.raycast(tasks)
    gpu_s = (time.perf_counter() - t0) / reps
    scene = orc.OracleScene(fmt, world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
    cores = orc.lib().or_max_threads()
    exp = scene.picker(tasks, threads=cores)
    t0 = time.perf_counter()
    for _ in range(5):
        scene.picker(tasks, threads=cores)
    cpu_s = (time.perf_counter() - t0) / 5
    return {"case": label, "rays": int(tasks.size), "identical_to_oracle": bool(got.tobytes() == exp.tobytes()), "hits": int((exp["dst"] > 0).sum()),
            "vx_raycast_ms": round(gpu_s * 1e3, 3), "vx_raycast_Mrays_s": round(tasks.size / gpu_s / 1e6, 1),
            "oracle_ms": round(cpu_s * 1e3, 3), "oracle_Mrays_s": round(tasks.size / cpu_s / 1e6, 2), "oracle_threads": cores}


def main():
    fmt = vra.SVO_CSVO
    chunk = vra.Chunk(0, 0, 0, 5)
    for x in range(32):
        for z in range(32):
            top = 8 + (int(host.lib().vxh_scene_hash32(0x5EED0001, 0, x, z)) & 7)
            for y in range(top + 1):
                chunk.set_block(x, y, z, 1 if y == top else (2 if y + 3 >= top else 3))
    chunk.compact()
    small = vra.World(fmt)
    small.set_chunk((0, 0, 0), chunk)
    small.serialize()
    out = [measure(fmt, small, tasks_for((16, 24, -24), (16, 8, 16), 256, 100.0), "C1: 256x256 rays, one 32^3 chunk")]
    big = vra.World(fmt)
    st = big.build_heightfield(12)
    n = float(1 << 12)
    eye = (0.5 * n, st["h_max"] + 0.05 * n, 0.5 * n)
    out.append(measure(fmt, big, tasks_for(eye, (eye[0] + 0.6, eye[1] - 0.35, eye[2] + 0.7), 256, -1.0), "256x256 rays, depth-12 bench scene"))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
