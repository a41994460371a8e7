"""Where do the compositing kernels' lanes go?  On a scene with config 3's density (same Gaussians per pixel, smaller image)
the ORACLE's lists are used to count, per (Gaussian, tile) instance of the reference's lists: the pixels that pass the exact
test (sigma >= 0, alpha >= 1/255), and how many pixel groups of various shapes contain at least one such pixel — the floor of
"group visits" for a kernel whose wave covers one group — with and without pixel saturation (position < n_contrib).
Test infrastructure (imports the oracle); output committed as profiles/r02/group_visit_stats.txt."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gsr_pkg
pkg=gsr_pkg.load()
from oracle import oracle as orc
W,H=480,272
n=int(1_000_000*(W*H)/(1920*1080))
s=pkg.synthetic.make_scene(n,W,H,3,1003)
cam=orc.Camera(W,H,s.focal)
st=orc.forward(s.means,s.shs,s.opacities,s.scales,s.rotations,cam,3)
gx=(W+15)//16
ids=st.values_sorted.astype(np.int64)
D=len(ids)
tile=np.zeros(D,np.int64)
for t,(a,b) in enumerate(st.ranges): tile[a:b]=t
pos=np.arange(D)-st.ranges[tile,0]
m=st.means2d[ids]; con=st.conics[ids]; o=s.opacities[ids]
X0=(tile%gx)*16; Y0=(tile//gx)*16
nc=st.n_contrib.reshape(H,W)
res={}
tot=0
acc={k:0 for k in ['any','rows4','quad','half8x4','rows2','r4x8','px']}
live_acc=dict(acc)
B=20000
for b0 in range(0,D,B):
    sl=slice(b0,min(D,b0+B))
    px=X0[sl,None,None]+np.arange(16)[None,None,:]
    py=Y0[sl,None,None]+np.arange(16)[None,:,None]
    dx=m[sl,0,None,None]-px; dy=m[sl,1,None,None]-py
    sig=con[sl,1,None,None]*dx*dy+0.5*(con[sl,0,None,None]*dx*dx+con[sl,2,None,None]*dy*dy)
    al=np.minimum(0.99,o[sl,None,None]*np.exp(-sig))
    act=(sig>=0)&(al>=1/255)&(px<W)&(py<H)
    # also within last contributor
    pyc=np.minimum(py,H-1); pxc=np.minimum(px,W-1)
    live=act&(pos[sl,None,None]<nc[pyc,pxc])
    for name,A in (('geo',act),('live',live)):
        d=acc if name=='geo' else live_acc
        d['any']+=A.any((1,2)).sum()
        d['px']+=A.sum()
        d['rows4']+=A.reshape(-1,4,4,16).any((2,3)).sum()       # 16x4 strips
        d['rows2']+=A.reshape(-1,8,2,16).any((2,3)).sum()       # 16x2
        d['quad']+=A.reshape(-1,2,8,2,8).any((2,4)).sum()       # 8x8
        d['half8x4']+=A.reshape(-1,4,4,2,8).any((2,4)).sum()    # 8 wide x 4 tall
        d['r4x8']+=A.reshape(-1,2,8,4,4).any((2,4)).sum()       # 4 wide x 8 tall
print("instances",D)
for name,d in (('geometric',acc),('live (before pixel saturation)',live_acc)):
    print(name)
    print("  instances with any active px: %.3f"%(d['any']/D))
    print("  active px per instance: %.1f"%(d['px']/D))
    for k,sz in (('rows4',64),('quad',64),('half8x4',32),('r4x8',32),('rows2',32)):
        print(f"  {k:8s}: groups/instance {d[k]/D:.3f}  lane efficiency {d['px']/(d[k]*sz):.3f}  lane-slots/instance {d[k]*sz/D:.1f}")
