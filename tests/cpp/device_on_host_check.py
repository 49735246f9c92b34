import sys, json, ctypes as C, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from helpers import oracle_scene, orc, golden_textures, golden_materials, vra
g=json.load(open('/root/repo/tests/golden/svo_shader_tests.json'))
L=C.CDLL('/tmp/libdevhost.so')
NORMALS=[[-1,0,0],[1,0,0],[0,-1,0],[0,1,0],[0,0,-1],[0,0,1]]
def expected(scene,tasks,ct):
    n=len(tasks); exp=np.zeros(n,dtype=orc.PICKER_RESULT_DTYPE)
    for i in range(n):
        r,_,_=scene.intersect(tasks[i]['pos'],tasks[i]['dir'],float(tasks[i]['max_dst']),bool(ct))
        if r.t>0:
            exp[i]['dst']=r.t; exp[i]['inside_voxel']=r.inside_voxel; exp[i]['pos']=list(r.pos); exp[i]['normal']=NORMALS[r.face_id]
        else: exp[i]['dst']=-1
    return exp
def run(fmt, scene, world, tasks, ct, image=False):
    n=len(tasks); exp=expected(scene,tasks,ct)
    frame=world.frame(pad_words=0)
    tex,mips=golden_textures(g); mats=golden_materials(g)
    out=np.zeros(n,dtype=orc.PICKER_RESULT_DTYPE)
    if image:
        nfb=C.c_uint32(0)
        L.devhost_picker_image(frame.ctypes.data_as(C.c_void_p), C.c_uint64(frame.size*4), mats.ctypes.data_as(C.c_void_p), mats.size, tex.ctypes.data_as(C.c_void_p), 4,4,4, tasks.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p), ct, C.byref(nfb))
    else:
        lo=(C.c_uint32*16)(0)
        L.devhost_picker(1 if fmt=='esvo' else 2, frame.ctypes.data_as(C.c_void_p), C.c_uint64(frame.size*4), mats.ctypes.data_as(C.c_void_p), mats.size, tex.ctypes.data_as(C.c_void_p), 4,4,4,1, lo, tasks.ctypes.data_as(C.c_void_p), n, out.ctypes.data_as(C.c_void_p), ct)
    return sum(out[i].tobytes()!=exp[i].tobytes() for i in range(n)), int((exp['dst']>0).sum())
ok=True
for ct in (0,1):
  for fmt in ('esvo','csvo'):
    for seed,svo_pos,n_blocks in ((2,(1,0,1),600),(3,(3,2,1),6000)):
        rng=np.random.default_rng(seed)
        pts=rng.integers(0,32,size=(n_blocks,3))
        blocks=[[int(x),int(y),int(z),int(rng.choice([1,2,3,4]))] for x,y,z in pts]
        scene,world=oracle_scene(g,fmt,svo_pos,blocks)
        n=3000
        tasks=np.zeros(n,dtype=orc.PICKER_TASK_DTYPE)
        tasks['pos']=(np.asarray(svo_pos,dtype=np.float32)*32+rng.uniform(-8,40,size=(n,3))).astype(np.float32)
        d=rng.normal(size=(n,3)); d/=np.linalg.norm(d,axis=1,keepdims=True)
        d[::13]=np.eye(3)[rng.integers(0,3,size=len(d[::13]))]
        tasks['dir']=d.astype(np.float32); tasks['max_dst']=np.where(rng.random(n)<0.3, rng.uniform(1,40,size=n), -1).astype(np.float32)
        r=run(fmt,scene,world,tasks,ct); print('ct',ct,fmt,n_blocks,r); ok&=r[0]==0
        if fmt=='esvo':
            r=run(fmt,scene,world,tasks,ct,image=True); print('   image',r); ok&=r[0]==0
w=vra.World(2); st=w.build_heightfield(8,threads=4)
sc=orc.OracleScene(2,w.frame(),golden_materials(g),*golden_textures(g))
rng=np.random.default_rng(5); n=3000
tasks=np.zeros(n,dtype=orc.PICKER_TASK_DTYPE)
tasks['pos']=rng.uniform(0,256,size=(n,3)).astype(np.float32)*np.float32([1,0.5,1])
d=rng.normal(size=(n,3)); d/=np.linalg.norm(d,axis=1,keepdims=True); tasks['dir']=d.astype(np.float32); tasks['max_dst']=-1
r=run('csvo',sc,w,tasks,1); print('heightfield csvo',r); ok&=r[0]==0
print('ALL OK' if ok else 'MISMATCH')
