#!/usr/bin/env python3
"""Ceiling of a beam start: primary rays of a sub-tile start at t_start = f * (the sub-tile's nearest hit); trips = descent to the cell there + what is left."""
import sys, ctypes as C
from pathlib import Path
import numpy as np
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[3]))
from _pkg import load_package
vra = load_package()
from oracle import oracle as orc
from voxel_rs_amd import scenes
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 12
W, H = (1920, 1080) if depth == 12 else (3840, 2160)
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 200
world = vra.World(vra.SVO_ESVO)
st = world.build_heightfield(depth)
tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
scene = orc.OracleScene(vra.SVO_ESVO, world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
u = scenes.bench_camera(depth, st["h_max"], W, H, shadow_distance=3.0e38)
U = orc.Uniforms.from_buffer_copy(bytes(u))
light = -np.array(list(U.light_dir), dtype=np.float32)
rng = np.random.default_rng(1)
FN = {0: (-1, 0, 0), 1: (1, 0, 0), 2: (0, -1, 0), 3: (0, 1, 0), 4: (0, 0, -1), 5: (0, 0, 1)}
lib = orc.lib()
base_p = base_s = 0
res_f = {0.999: 0, 0.99: 0, 0.95: 0, 0.9: 0, 0.8: 0}
nt = 0
for _ in range(NS):
    bx, by = int(rng.integers(0, W // 8)) * 8, int(rng.integers(0, H // 8)) * 8
    rays = []
    smax = 0
    for y in range(by, by + 8):
        for x in range(bx, bx + 8):
            ro = (C.c_float * 3)(); rd = (C.c_float * 3)()
            lib.or_primary_ray(C.byref(U), W, H, x, y, C.byref(ro), C.byref(rd))
            res, fr, n = scene.intersect(list(ro), list(rd), -1.0, 1, max_frames=600)
            rays.append((n, fr["t_min"].copy(), fr["scale"].astype(int).copy(), res.t))
            if res.t >= 0:
                nrm = np.array(FN[res.face_id], dtype=np.float32)
                so = np.array(list(res.pos), dtype=np.float32) + nrm * np.float32(0.001)
                r2, f2, n2 = scene.intersect(list(so), list(light), -1.0, 1, max_frames=2)
                smax = max(smax, n2)
    nt += 1
    mp = max(n for n, *_ in rays)
    base_p += mp; base_s += smax
    hits = [t for *_, t in rays if t >= 0]
    for f in res_f:
        if not hits:
            res_f[f] += mp   # (a sky sub-tile: the beam finds nothing; no gain counted, although its rays could be skipped altogether)
            continue
        ts = f * min(hits)
        worst = 0
        for n, tm, sc, t in rays:
            k = int(np.searchsorted(tm, ts, side="right"))  # iterations whose t_min <= ts: skipped, but for the descent to the cell at ts
            k = max(k - 1, 0)
            levels = 22 - sc[k] if k < len(sc) else 0
            worst = max(worst, min(n, (n - k) + levels))
        res_f[f] += worst
print("sub-tiles", nt, "primary trips", base_p / nt, "shadow trips", base_s / nt)
for f, v in res_f.items():
    print(f"t_start = {f} x nearest hit: primary trips {v / nt:.2f}  total {(v + base_s) / (base_p + base_s):.3f} of lockstep")
