#!/usr/bin/env python3
"""Shadow rays through a per-wave queue: a wave traces the primary rays of S sub-tiles in lockstep, queues their shadow rays, then traces those 64 at a time and
refills idle lanes from the queue when at least R are idle. Trips and refill events against lockstep (oracle's iteration counts, C3 view 7)."""
import sys
import numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from _pkg import load_package
vra = load_package()
from oracle import oracle as orc
from voxel_rs_amd import scenes
import bench
depth = 12; W, H = 1920, 1080
world = vra.World(vra.SVO_ESVO)
st = world.build_heightfield(depth)
tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
scene = orc.OracleScene(vra.SVO_ESVO, world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
u = bench.moving_uniforms(scenes, depth, st["h_max"], W, H, 7)
U = orc.Uniforms.from_buffer_copy(bytes(u))
hits = scene.render(U, W, H)[1]
U0 = orc.Uniforms.from_buffer_copy(bytes(u)); U0.render_shadows = 0
prim = scene.render(U0, W, H)[1]["steps"].astype(np.int64)
sh = np.where((hits["flags"] & 2) != 0, hits["steps"].astype(np.int64) - prim, 0)
def batches(a): return a.reshape(H // 8, 8, W // 8, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
P, S = batches(prim), batches(sh)
n = len(P)
waves = 4096
rng = np.random.default_rng(0)
order = rng.permutation(n)  # which wave gets which sub-tile depends on timing: random is close enough
per_wave = [order[w::waves] for w in range(waves)]
lock_p = int(P.max(1).sum()); lock_s = int(S.max(1).sum())
print("lockstep: primary trips", lock_p, "shadow trips", lock_s)
def session(queue, R):
    queue = list(queue)
    lanes = np.zeros(64, np.int64)
    trips = events = 0
    qi = 0
    while True:
        idle = lanes <= 0
        if qi < len(queue) and (idle.sum() >= R or idle.all()):
            k = min(int(idle.sum()), len(queue) - qi)
            idx = np.flatnonzero(idle)[:k]
            lanes[idx] = queue[qi:qi + k]; qi += k
            events += 1
            continue
        if (lanes > 0).sum() == 0:
            break
        # run until the next moment a refill could happen (or the end): step to the smallest remaining count that changes the idle count enough
        act = lanes[lanes > 0]
        if qi >= len(queue):
            step = int(act.max())
        else:
            need = R - int(idle.sum())
            step = int(np.sort(act)[min(need, len(act)) - 1])
        lanes -= step
        trips += step
    return trips, events
for Ssz in (1, 2, 4, 8):
    for R in (64, 48, 32, 16):
        trips = events = 0
        for mine in per_wave:
            for i in range(0, len(mine), Ssz):
                q = np.concatenate([S[j][S[j] > 0] for j in mine[i:i + Ssz]]) if len(mine[i:i + Ssz]) else np.zeros(0, np.int64)
                if len(q) == 0: continue
                t, e = session(q, R)
                trips += t; events += e
        print(f"session of {Ssz} sub-tiles, refill at {R} idle lanes: shadow trips {trips} ({trips / lock_s:.3f} of lockstep), events {events}; frame trips {(lock_p + trips) / (lock_p + lock_s):.3f}")
