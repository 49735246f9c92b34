import numpy as np
P=np.load('/tmp/steps_primary.npy'); T=np.load('/tmp/steps_total.npy'); S=T-P
H,W=P.shape
Hp=(H+31)//32*32; Wp=(W+31)//32*32
def pad(a):
    b=np.zeros((Hp,Wp),dtype=a.dtype); b[:H,:W]=a; return b
P=pad(P); S=pad(S)
def blocks(a,bh,bw):
    return a.reshape(Hp//bh,bh,Wp//bw,bw).transpose(0,2,1,3).reshape(-1,bh,bw)
p8=blocks(P,8,8).reshape(-1,64); s8=blocks(S,8,8).reshape(-1,64)
base=p8.max(1).sum()+s8.max(1).sum()
print('base trips',base)
for (gh,gw) in ((1,1),(1,2),(1,4),(2,2),(2,4),(1,8)):
    pb=blocks(P,16,16); sb=blocks(S,16,16)   # [nb,16,16]
    nb=pb.shape[0]
    # groups inside block
    pg=pb.reshape(nb,16//gh,gh,16//gw,gw).transpose(0,1,3,2,4).reshape(nb,-1,gh*gw)
    sg=sb.reshape(nb,16//gh,gh,16//gw,gw).transpose(0,1,3,2,4).reshape(nb,-1,gh*gw)
    key=(pg+sg).max(2)   # group cost = max total of its pixels
    idx=np.argsort(key,axis=1,kind='stable')
    ng=key.shape[1]; per=ng//4
    pmax=np.take_along_axis(pg.max(2),idx,1).reshape(nb,4,per).max(2).sum()
    smax=np.take_along_axis(sg.max(2),idx,1).reshape(nb,4,per).max(2).sum()
    print(f'groups {gh}x{gw}: trips ratio {(pmax+smax)/base:.3f}')
