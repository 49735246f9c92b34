import sys, ctypes as C
import numpy as np
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[3]))
from _pkg import load_package
vra = load_package()
from oracle import oracle as orc
from voxel_rs_amd import scenes
depth=12; W,H=1920,1080
world = vra.World(vra.SVO_ESVO)
st = world.build_heightfield(depth)
tex, mats = scenes.synthetic_textures(), scenes.synthetic_materials()
scene = orc.OracleScene(vra.SVO_ESVO, world.frame(), mats.view(orc.MATERIAL_DTYPE), tex, 6)
u = scenes.bench_camera(depth, st["h_max"], W, H, shadow_distance=3.0e38)
hits = scene.render(orc.Uniforms.from_buffer_copy(bytes(u)), W, H)[1]
u0 = scenes.bench_camera(depth, st["h_max"], W, H, shadow_distance=3.0e38, render_shadows=False)
prim = scene.render(orc.Uniforms.from_buffer_copy(bytes(u0)), W, H)[1]["steps"].astype(np.int64)
sh = np.where((hits["flags"] & 2) != 0, hits["steps"].astype(np.int64) - prim, 0)
hit = (hits["flags"] & 1) != 0
occl = (hits["flags"] & 4) != 0
print("pixels", W*H, "hit", hit.mean(), "shadow rays occluded", occl.sum()/max(1,((hits["flags"]&2)!=0).sum()))
P = prim.reshape(H//8, 8, W//8, 8).transpose(0,2,1,3).reshape(-1, 64)
S = sh.reshape(H//8, 8, W//8, 8).transpose(0,2,1,3).reshape(-1, 64)
print("primary: mean iters %.1f, batch max mean %.1f; sum of batch max %d" % (prim.mean(), P.max(1).mean(), P.max(1).sum()))
hasS = S.max(1) > 0
print("shadow: rays %d mean iters %.1f, batches %d of %d, batch max mean %.1f; sum %d" % ((sh>0).sum(), sh[sh>0].mean(), hasS.sum(), len(S), S[hasS].max(1).mean(), S.max(1).sum()))
skyb = ~hasS
print("sky batches: primary batch max mean %.1f (sum %d); ground batches primary max mean %.1f" % (P[skyb].max(1).mean(), P[skyb].max(1).sum(), P[hasS].max(1).mean()))
print("lane use primary %.3f shadow %.3f" % (prim.sum()/(P.max(1).sum()*64), sh.sum()/(S.max(1).sum()*64)))
so = sh[occl]; sn = sh[(sh>0)&~occl]
print("occluded shadow rays mean %.1f, unoccluded mean %.1f" % (so.mean(), sn.mean()))
print("shadow iters percentiles", np.percentile(sh[sh>0],[10,50,90,99,100]))
print("primary iters percentiles", np.percentile(prim,[10,50,90,99,100]))
