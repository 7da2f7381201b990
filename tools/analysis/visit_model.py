"""tools/analysis/visit_model.py -- ANALYSIS TOOLING (CPU, uses the oracle; not part of the product path).

Where do the rasterizer's lanes go?  For one view of BASELINE config 2 (bench.make_inputs) the oracle's depth-sorted tile
lists and per-pixel final indices give, for every candidate wave footprint (16x16 ... 4x8 pixels):
  * visits            (entry, footprint) pairs that must be walked: the footprint's longest pixel walk, restricted to
                      entries with at least one contributing pixel in the footprint (= perfect per-footprint culling)
  * useful lanes      contributing (entry, pixel) pairs / (visits x footprint pixels)
and the length histogram of the tile lists.  Printed numbers are quoted in DESIGN.md section 9 (profiles/r02k_visit_model.txt).
"""
import sys, torch, numpy as np
sys.path.insert(0,'/root/repo')
import bench
from oracle import cref, shade_ref
cfg=dict(bench.CFG, views_per_gpu=1)
t=bench.make_inputs(cfg,"cpu",rank=0)
H,W=cfg["height"],cfg["width"]
cref.set_threads(32)
with torch.no_grad():
    preds=shade_ref.shade(t["f_vn"],t["f_vc"],t["postex"],t["tn"],t["albedo"],t["light_sh"],t["campos"],envmips=t["mips"],lightrot=t["lightrot"])
means,scales,quats=preds["primpos"][0],preds["primscale"][0],preds["primqvec"][0]
K,vm=t["K"][0],t["Rt"][0]
fx,fy,cx,cy=float(K[0,0]),float(K[1,1]),float(K[0,2]),float(K[1,2])
xys,depths,radii,conics,comp,nth,cov3d=cref.project_gaussians(means,scales,1.0,quats,vm,fx,fy,cx,cy,H,W,16,0.1)
_,ids,bins=cref.bin_and_sort(xys,depths,radii,nth,H,W,16)
opac=(preds["opacity"][0,:,0]*comp).contiguous()
col4=torch.cat([preds["color"][0],depths[:,None]],1).contiguous()
img,Ts,idx=cref.rasterize_forward(ids,bins,xys,conics,col4,opac,H,W,16,torch.zeros(4))
tx,ty=(W+15)//16,(H+15)//16
bins=bins.reshape(ty,tx,2).numpy()
idx=idx.numpy(); Ts=Ts.numpy()
Hp,Wp=ty*16,tx*16
walk=np.zeros((Hp,Wp),np.int64)
start=np.repeat(np.repeat(bins[...,0],16,0),16,1)[:H,:W]
n=np.where(Ts<1.0, idx-start+1, 0)
walk[:H,:W]=n
print("isect",ids.numel(),"mean walk per pixel",walk[:H,:W].mean())
def agg(bh,bw):
    w=walk.reshape(Hp//bh,bh,Wp//bw,bw).max((1,3))
    return w.sum()*bh*bw, w.sum()
tot=walk.sum()
for bh,bw in ((16,16),(8,16),(8,8),(4,16),(4,8),(4,4)):
    lanes,vis=agg(bh,bw)
    print(f"block {bh}x{bw}: visits {vis}  lane-visits {lanes}  useful fraction {tot/lanes:.3f}")
# useful-lane fraction: contributions (alpha>=1/255, before termination) per (entry, block) visit
rng=np.random.default_rng(0)
xy=xys.numpy(); con=conics.numpy(); op=opac.numpy(); idsn=ids.numpy()
tiles=[(a,b) for a in range(ty) for b in range(tx) if bins[a,b,1]>bins[a,b,0]]
sel=rng.choice(len(tiles),1500,replace=False)
stats={k:[0,0,0] for k in ((16,16),(8,16),(8,8),(4,16),(4,8))}  # visits, lane-visits, contributions
for si in sel:
    a,b=tiles[si]; s,e=bins[a,b]
    y0,x0=a*16,b*16
    yy,xx=np.mgrid[y0:y0+16,x0:x0+16]
    inside=(yy<H)&(xx<W)
    fin=np.where(inside, np.where(walk[np.minimum(yy,Hp-1),np.minimum(xx,Wp-1)]>0, walk[np.minimum(yy,Hp-1),np.minimum(xx,Wp-1)],0),0)  # entries walked per pixel
    m=int(fin.max())
    if m==0: continue
    g=idsn[s:s+m]
    dx=xy[g,0][:,None,None]-(xx[None]+0.5); dy=xy[g,1][:,None,None]-(yy[None]+0.5)
    sig=0.5*(con[g,0][:,None,None]*dx*dx+con[g,2][:,None,None]*dy*dy)+con[g,1][:,None,None]*dx*dy
    al=np.minimum(0.999,op[g][:,None,None]*np.exp(-sig))
    contrib=(sig>=0)&(al>=1/255)&(np.arange(m)[:,None,None]<fin[None])
    for (bh,bw),st in stats.items():
        c=contrib.reshape(m,16//bh,bh,16//bw,bw).sum((2,4))   # per entry per block: contributing pixels
        f=fin.reshape(16//bh,bh,16//bw,bw).max((1,3))          # block walk length
        alive=(np.arange(m)[:,None,None]<f[None])
        vis_all=alive.sum()                     # visits without ellipse culling
        vis=(alive&(c>0)).sum()                 # visits with perfect per-block culling
        st[0]+=vis; st[1]+=vis*bh*bw; st[2]+=contrib.sum()
        st.append(vis_all) if len(st)==3 else st.__setitem__(3,st[3]+vis_all)
for k,st in stats.items():
    print(f"block {k}: visits(any contributor) {st[0]}  of walked {st[3]}  lane-visits {st[1]}  contributions {st[2]}  useful lanes {st[2]/st[1]:.3f}")
ln=(bins[...,1]-bins[...,0]).reshape(-1)
print("tiles",ln.size,"empty",(ln==0).sum())
for lo,hi in ((1,64),(65,128),(129,256),(257,512),(513,1024),(1025,2048),(2049,4096),(4097,1<<30)):
    m=(ln>=lo)&(ln<=hi); print(f"len {lo}-{hi}: tiles {m.sum()} entries {ln[m].sum()} ({ln[m].sum()/ln.sum():.3f})")
