#!/usr/bin/env python3
"""Trips saved if a ray ended (miss) when it leaves the bounding box of the world's solid content, and started where it enters it."""
import sys, ctypes as C
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
print(st)
N = 1 << depth
hmax = float(st["h_max"]) + 1.0
tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
scene = orc.OracleScene(vra.SVO_ESVO, world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
u = scenes.bench_camera(depth, st["h_max"], W, H, shadow_distance=3.0e38)
U = orc.Uniforms.from_buffer_copy(bytes(u))
light = -np.array(list(U.light_dir), dtype=np.float32)
rng = np.random.default_rng(1)
FN = {0: (-1, 0, 0), 1: (1, 0, 0), 2: (0, -1, 0), 3: (0, 1, 0), 4: (0, 0, -1), 5: (0, 0, 1)}
lib = orc.lib()
def span(ro, rd):
    lo, hi = 0.0, 1e30
    for a, (bl, bh) in enumerate(((0, N), (0, hmax), (0, N))):
        d = rd[a] if abs(rd[a]) > 1e-12 else 1e-12
        t0, t1 = (bl - ro[a]) / d, (bh - ro[a]) / d
        lo, hi = max(lo, min(t0, t1)), min(hi, max(t0, t1))
    return lo, hi
tot = dict(p=0, s=0, p_exit=0, s_exit=0, p_both=0)
sky = 0
nt = 0
for _ in range(NS):
    bx, by = int(rng.integers(0, W // 8)) * 8, int(rng.integers(0, H // 8)) * 8
    P, S, Pe, Se, Pb = [0], [0], [0], [0], [0]
    for y in range(by, by + 8):
        for x in range(bx, bx + 8):
            ro = (C.c_float * 3)(); rd = (C.c_float * 3)()
            lib.or_primary_ray(C.byref(U), W, H, x, y, C.byref(ro), C.byref(rd))
            res, fr, n = scene.intersect(list(ro), list(rd), -1.0, 1, max_frames=600)
            lo, hi = span(list(ro), list(rd))
            tm = fr["t_min"]
            if hi <= lo:
                ne = nb = 0
            else:
                ne = int(np.searchsorted(tm, hi, side="right"))   # iterations that begin inside the box
                ne = min(n, ne + 0)
                k = max(int(np.searchsorted(tm, lo, side="right")) - 1, 0)
                nb = min(n, ne - k + (22 - int(fr["scale"][k])))
            P.append(n); Pe.append(ne); Pb.append(nb)
            if res.t >= 0:
                nrm = np.array(FN[res.face_id], dtype=np.float32)
                so = np.array(list(res.pos), dtype=np.float32) + nrm * np.float32(0.001)
                r2, f2, n2 = scene.intersect(list(so), list(light), -1.0, 1, max_frames=600)
                lo2, hi2 = span(list(so), list(light))
                S.append(n2); Se.append(min(n2, int(np.searchsorted(f2["t_min"], hi2, side="right"))))
    nt += 1
    if max(S) == 0: sky += 1
    tot["p"] += max(P); tot["s"] += max(S); tot["p_exit"] += max(Pe); tot["s_exit"] += max(Se); tot["p_both"] += max(Pb)
b = tot["p"] + tot["s"]
print("sub-tiles", nt, "without a hit", sky, {k: round(v / nt, 2) for k, v in tot.items()})
print("end at the box's far side: %.3f of lockstep; + start at its near side: %.3f" % ((tot["p_exit"] + tot["s_exit"]) / b, (tot["p_both"] + tot["s_exit"]) / b))
