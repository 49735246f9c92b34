import numpy as np
P=np.load('/tmp/steps_primary.npy'); T=np.load('/tmp/steps_total.npy'); S=T-P
H,W=P.shape; print(P.shape, P.mean(), S.mean(), (S>0).mean())
# pad to multiple of 32
Hp=(H+31)//32*32; Wp=(W+31)//32*32
def pad(a):
    b=np.zeros((Hp,Wp),dtype=a.dtype); b[:H,:W]=a; return b
P=pad(P); S=pad(S)
def blocks(a,bh,bw):
    return a.reshape(Hp//bh,bh,Wp//bw,bw).transpose(0,2,1,3).reshape(-1,bh*bw)
def cost_unsorted():
    p=blocks(P,8,8); s=blocks(S,8,8)
    return p.max(1).sum()+s.max(1).sum(), p.sum()+s.sum()
base,work=cost_unsorted()
print('current 8x8: trips',base,'lane-slot utilisation',work/(base*64))
def cost_sorted(bh,bw,key='total',sep=False):
    p=blocks(P,bh,bw); s=blocks(S,bh,bw); n=bh*bw; g=n//64
    tot=0
    if not sep:
        k = (p+s) if key=='total' else (p if key=='primary' else np.maximum(p,s))
        idx=np.argsort(k,axis=1,kind='stable')
        ps=np.take_along_axis(p,idx,1).reshape(-1,g,64); ss=np.take_along_axis(s,idx,1).reshape(-1,g,64)
        tot=ps.max(2).sum()+ss.max(2).sum()
    else:
        ps=np.sort(p,axis=1).reshape(-1,g,64); ss=np.sort(s,axis=1).reshape(-1,g,64)
        tot=ps.max(2).sum()+ss.max(2).sum()
    return tot
for (bh,bw) in ((8,16),(16,16),(16,32),(32,32)):
    for key in ('total','primary','max'):
        c=cost_sorted(bh,bw,key); print(f'sorted {bh}x{bw} by {key}: trips {c}  ratio {c/base:.3f}')
    c=cost_sorted(bh,bw,sep=True); print(f'sorted {bh}x{bw} separately (lower bound): ratio {c/base:.3f}')
