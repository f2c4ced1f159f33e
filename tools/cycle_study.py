#!/usr/bin/env python3
"""CPU study behind the adaptive cycle detector: when do the fp64 orbits of interior pixels of the C4 view
become EXACTLY periodic, with which period, and what would an ideal detector save?  numpy, same operation order as the kernels."""
import numpy as np, time
# C4 view: 8192x8192 frame, sample a coarse grid of pixels with the frame's exact pixel mapping (mandelbrot.comp:149-151)
W=H=8192; cx0,cy0,zoom=-0.743643887037151, 0.13182590420533, 1e-6; MI=16384
n=40
xs=(np.arange(n)*(W//n)+7).astype(np.float64); ys=(np.arange(n)*(H//n)+3).astype(np.float64)
PX,PY=np.meshgrid(xs,ys)
uvx=(PX-0.5*W)/H; uvy=(PY-0.5*H)/H
cx=(cx0+uvx*zoom).ravel(); cy=(cy0+uvy*zoom).ravel()
N=cx.size
zx=np.zeros(N); zy=np.zeros(N)
orbx=np.empty((MI+1,N)); orby=np.empty((MI+1,N))
esc=np.full(N,MI,dtype=np.int64); alive=np.ones(N,bool)
t0=time.time()
with np.errstate(all='ignore'):
    for i in range(MI):
        orbx[i]=zx; orby[i]=zy
        x=(zx*zx-zy*zy)+cx; y=((2.0*zx)*zy)+cy
        zx,zy=x,y
        e=alive&((zx*zx+zy*zy)>16.0)
        esc[e]=i; alive&=~e
    orbx[MI]=zx; orby[MI]=zy
print("iterated",time.time()-t0,"s; interior frac",alive.mean())
idx=np.where(alive)[0]
res=[]
for j in idx:
    ox,oy=orbx[:,j],orby[:,j]
    # period: smallest p with z_MI == z_{MI-p}
    eq=(ox[:MI]==ox[MI])&(oy[:MI]==oy[MI])
    w=np.where(eq)[0]
    if w.size==0:
        res.append((-1,-1)); continue
    p=MI-w[-1]
    # preperiod: first n with z_n == z_{n+p}
    m=(ox[:MI+1-p]==ox[p:])&(oy[:MI+1-p]==oy[p:])
    c=int(np.argmax(m)) if m.any() else -1
    res.append((p,c))
res=np.array(res)
never=(res[:,0]<0).mean()
print("interior pixels",len(idx),"never exactly periodic within 16384:",never)
ok=res[res[:,0]>0]
print("periods:",np.unique(ok[:,0],return_counts=True))
c=ok[:,1]
print("exact preperiod c: min %d p10 %d p50 %d p90 %d max %d mean %.0f"%(c.min(),np.percentile(c,10),np.percentile(c,50),np.percentile(c,90),c.max(),c.mean()))
# what the current detector achieves: first block boundary (multiple of 16 after b0=192?) where snapshot logic hits: approximate ideal: c + p
ideal=(c+ok[:,0]).mean()
print("ideal mean closing time (c+p) over periodic interior pixels: %.0f ; fraction of 16384: %.2f"%(ideal, ideal/MI))
tot_ideal=(np.where(res[:,0]>0, res[:,1]+res[:,0], MI)).mean()
print("mean executed iterations over interior pixels with an ideal detector: %.0f (vs 16384)"%tot_ideal)
