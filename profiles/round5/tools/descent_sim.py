#!/usr/bin/env python3
"""How many trips of the lockstep loop are a ray's initial descent (consecutive PUSHes from the root), and what skipping them would save."""
import sys, ctypes as C
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[3]
sys.path.insert(0, str(ROOT))
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
def descent(frames):
    s = frames["scale"].astype(int)
    d = 0
    while d + 1 < len(s) and s[d + 1] == s[d] - 1:
        d += 1
    return d
tot = dict(base=0, pyr4=0, pyr6=0, pyr8=0, chain=0, chain_pyr6=0, all_=0)
dps, dss, dcs = [], [], []
n_tiles = 0
for _ in range(NS):
    bx, by = int(rng.integers(0, W // 8)) * 8, int(rng.integers(0, H // 8)) * 8
    P, S = [], []
    for y in range(by, by + 8):
        for x in range(bx, bx + 8):
            ro = (C.c_float * 3)(); rd = (C.c_float * 3)()
            lib.or_primary_ray(C.byref(U), W, H, x, y, C.byref(ro), C.byref(rd))
            res, fr, n = scene.intersect(list(ro), list(rd), -1.0, 1, max_frames=600)
            dp = descent(fr)
            P.append((n, dp))
            if res.t >= 0:
                nrm = np.array(FN[res.face_id], dtype=np.float32)
                pos = np.array(list(res.pos), dtype=np.float32)
                so = pos + nrm * np.float32(0.001)
                r2, f2, n2 = scene.intersect(list(so), list(light), -1.0, 1, max_frames=600)
                ds = descent(f2)
                v = np.floor(pos - nrm * np.float32(0.001)).astype(np.int64)
                c = np.floor(so).astype(np.int64)
                x_ = int(np.max(v ^ c))
                dc = depth - x_.bit_length()
                S.append((n2, ds, dc))
                dss.append(ds); dcs.append(dc)
            dps.append(dp)
    n_tiles += 1
    mp = max(n for n, _ in P)
    ms = max((n for n, _, _ in S), default=0)
    tot["base"] += mp + ms
    for k in (4, 6, 8):
        tot[f"pyr{k}"] += max(n - min(d, k) for n, d in P) + max((n - min(d, k) for n, d, _ in S), default=0)
    tot["chain"] += mp + max((n - min(d, dc) for n, d, dc in S), default=0)
    tot["chain_pyr6"] += max(n - min(d, 6) for n, d in P) + max((n - max(min(d, dc), min(d, 6)) for n, d, dc in S), default=0)
    tot["all_"] += max(n - d for n, d in P) + max((n - d for n, d, _ in S), default=0)
print("sub-tiles", n_tiles, "trips per sub-tile", {k: round(v / n_tiles, 2) for k, v in tot.items()})
print("relative", {k: round(v / tot["base"], 3) for k, v in tot.items()})
print("primary descent: mean %.2f hist %s" % (np.mean(dps), np.bincount(dps).tolist()))
print("shadow descent: mean %.2f hist %s" % (np.mean(dss), np.bincount(dss).tolist()))
print("shadow common with primary: mean %.2f hist %s" % (np.mean(dcs), np.bincount(np.maximum(dcs, 0)).tolist()))
