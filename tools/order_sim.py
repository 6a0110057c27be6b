"""CPU simulation (numpy) of how many supergroup / group / half spheres a line reaches per cloud under
different orderings of the records: the kernel's Hilbert cell order, a local k-d reordering inside every
supergroup of 64 (or block of 256), and a full k-d order.  The culled scan is VALU-issue bound, so these
counts translate into its instruction count.  Needs no GPU.  (Uses the product's sampler inputs only
through synth; lines are random chords drawn here.)"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
from rrl_hip import synth
def hilbert_cell(q0,q1,q2):
    x=[q0,q1,q2]
    q=8
    while q>1:
        p=q-1
        for i in range(3):
            setb = -((x[i]//q)&1) & 0xffffffff
            t=(x[0]^x[i]) & p & ~setb
            x[0]^= (p & setb) | t
            x[i]^= t
        q>>=1
    x[1]^=x[0]; x[2]^=x[1]
    t=0; q=8
    while q>1:
        t ^= (q-1) & (-((x[2]//q)&1) & 0xffffffff)
        q>>=1
    key=0
    for bit in range(3,-1,-1):
        for i in range(3):
            key=(key<<1)|(((x[i]^t)>>bit)&1)
    return key
LUT=np.array([[[hilbert_cell(a,b,c) for c in range(16)] for b in range(16)] for a in range(16)])
def order_hilbert(P):
    mn=P.min(0); ext=P.max(0)-mn
    q=np.clip(((P-mn)*(15.999/ext)).astype(int),0,15)
    key=LUT[q[:,0],q[:,1],q[:,2]]
    return np.argsort(key,kind='stable')
def spheres(P,thr,size):
    n=len(P); k=n//size
    Pg=P.reshape(k,size,3); lo=Pg.min(1); hi=Pg.max(1); c=0.5*lo+0.5*hi
    rho=np.linalg.norm(Pg-c[:,None],axis=2).max(1)
    R=rho+thr.reshape(k,size).max(1)
    return c,R
def count(P,thr,lines):
    d=lines[:,:3]; o=lines[:,3:]
    res=[]
    prev=None
    for size in (64,16,8):
        c,R=spheres(P,thr,size)
        a=c[None]-o[:,None]          # (L,k,3)
        dot=(a*d[:,None]).sum(-1)
        d2=(a*a).sum(-1)-dot*dot
        ok=d2<=R[None]**2
        if prev is not None:
            rep=prev.shape[1]
            ok &= np.repeat(prev, ok.shape[1]//rep, axis=1)
        res.append(ok.sum(1).mean()); prev=ok
    # exact point-0 passes
    return res
def local_kd(P, idx, size_leaf=8):
    # reorder indices within the block by recursive median split on the widest axis down to leaves
    if len(idx)<=size_leaf: return idx
    pts=P[idx]; ax=np.argmax(pts.max(0)-pts.min(0))
    o=np.argsort(pts[:,ax],kind='stable'); h=len(idx)//2
    return np.concatenate([local_kd(P,idx[o[:h]],size_leaf), local_kd(P,idx[o[h:]],size_leaf)])
pr=synth.make_pair(0,4096,4096)
rng=np.random.default_rng(0)
u=rng.standard_normal((2,4000,3)); u/=np.linalg.norm(u,axis=2,keepdims=True)
q1,q2=pr['radius']*u[0]+pr['center'],pr['radius']*u[1]+pr['center']
dd=q2-q1; lines=np.concatenate([dd/np.linalg.norm(dd,axis=1,keepdims=True),q1],1)
def thresholds(tri):
    p=tri.reshape(-1,3,3).astype(np.float64)
    e=(np.linalg.norm(p[:,1]-p[:,0],axis=1)+np.linalg.norm(p[:,2]-p[:,0],axis=1)+np.linalg.norm(p[:,1]-p[:,2],axis=1))/3
    return e*1.731/2
for name in ('src_tri','tar_tri'):
    tri=pr[name]; P=tri[:,:3].astype(np.float64); thr=thresholds(tri)
    o=order_hilbert(P)
    base=count(P[o],thr[o],lines)
    # local k-d inside each supergroup of 64
    o2=np.concatenate([local_kd(P,o[i:i+64]) for i in range(0,4096,64)])
    loc=count(P[o2],thr[o2],lines)
    # local k-d inside blocks of 256 (4 supergroups)
    o3=np.concatenate([local_kd(P,o[i:i+256]) for i in range(0,4096,256)])
    loc256=count(P[o3],thr[o3],lines)
    o4=local_kd(P,np.arange(4096))
    full=count(P[o4],thr[o4],lines)
    print(name,'hilbert sg/grp/half per line:',[round(x,2) for x in base],'| local k-d in 64:',[round(x,2) for x in loc],'| in 256:',[round(x,2) for x in loc256],'| full k-d:',[round(x,2) for x in full])


# ---- tighter spheres for the SAME (Hilbert) order: AABB centre (the kernel's) vs the better of AABB centre /
#      centroid / one Ritter step, per node
def spheres_best(P, thr, size):
    n = len(P); k = n // size
    Pg = P.reshape(k, size, 3); lo = Pg.min(1); hi = Pg.max(1)
    c0 = 0.5 * lo + 0.5 * hi
    r0 = np.linalg.norm(Pg - c0[:, None], axis=2).max(1)
    c1 = Pg.mean(1)
    r1 = np.linalg.norm(Pg - c1[:, None], axis=2).max(1)
    # Ritter: start at AABB centre sphere, grow towards the farthest point a few times from a smaller start
    c2 = c0.copy(); r2 = r0 * 0.85
    for _ in range(8):
        d = np.linalg.norm(Pg - c2[:, None], axis=2); j = d.argmax(1); dm = d.max(1)
        far = Pg[np.arange(k), j]
        grow = dm > r2
        newr = np.where(grow, 0.5 * (r2 + dm), r2)
        shift = np.where(grow, (dm - newr) / np.maximum(dm, 1e-30), 0.0)
        c2 = c2 + (far - c2) * shift[:, None]; r2 = newr
    r2 = np.linalg.norm(Pg - c2[:, None], axis=2).max(1)
    best = np.argmin(np.stack([r0, r1, r2]), 0)
    c = np.where((best == 0)[:, None], c0, np.where((best == 1)[:, None], c1, c2))
    r = np.minimum(np.minimum(r0, r1), r2)
    return c, r + thr.reshape(k, size).max(1), (r / r0).mean()

def count_best(P, thr, lines):
    d = lines[:, :3]; o = lines[:, 3:]
    res = []; prev = None; ratios = []
    for size in (64, 16, 8):
        c, R, ratio = spheres_best(P, thr, size)
        a = c[None] - o[:, None]
        dot = (a * d[:, None]).sum(-1)
        d2 = (a * a).sum(-1) - dot * dot
        ok = d2 <= R[None] ** 2
        if prev is not None:
            ok &= np.repeat(prev, ok.shape[1] // prev.shape[1], axis=1)
        res.append(round(float(ok.sum(1).mean()), 2)); prev = ok; ratios.append(round(float(ratio), 3))
    return res, ratios

for name in ('src_tri', 'tar_tri'):
    tri = pr[name]; P = tri[:, :3].astype(np.float64); thr = thresholds(tri)
    o = order_hilbert(P)
    print(name, 'hilbert order, best-of-three spheres: sg/grp/half per line', *count_best(P[o], thr[o], lines), '(mean rho / rho_AABB-centre per level)')
