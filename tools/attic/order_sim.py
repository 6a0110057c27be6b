"""CPU simulation (numpy) of the culled scan's executed work: sphere tests per tree level and exact point-0
tests per (line, cloud) under different orderings of the records -- the kernel's Hilbert cell order, a local
k-d reordering inside every supergroup of 64 / block of 256, a full k-d order -- and different level
structures.  With lines accepted like the sampler's (chords of the sampling sphere that hit both clouds'
boxes) the Hilbert row reproduces the kernel's own counters (rrl_scan_counters) at the bench shape:
64 / 32.4 / 20.6 / 88.4.  Needs no GPU.  Results: profiles/r02c_kd_refine_and_stream_split_experiments.txt."""
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


def count_levels(P, thr, lines, sizes):
    """tests per line at every level (top level: all nodes; below: fan-out x passing parents), last = exact tests"""
    d = lines[:, :3]; o = lines[:, 3:]
    tests = []; prev = None
    for size in sizes:
        c, R = spheres(P, thr, size)
        a = c[None] - o[:, None]; dot = (a * d[:, None]).sum(-1); d2 = (a * a).sum(-1) - dot * dot
        ok = d2 <= R[None] ** 2
        if prev is None:
            tests.append(ok.shape[1])
        else:
            par = np.repeat(prev, ok.shape[1] // prev.shape[1], axis=1)
            tests.append(par.sum(1).mean()); ok &= par
        prev = ok
    tests.append(prev.sum(1).mean() * sizes[-1])
    return [round(float(t), 1) for t in tests]

def kd_blocks(P, o, blk, leaf=8):
    return np.concatenate([local_kd(P, o[i:i + blk], leaf) for i in range(0, len(o), blk)])

def hits_box(lines, P):
    lo = P.min(0); hi = P.max(0); d = lines[:, :3]; o = lines[:, 3:]
    with np.errstate(divide='ignore', invalid='ignore'):
        t1 = (lo - o) / d; t2 = (hi - o) / d
    return np.minimum(t1, t2).max(1) <= np.maximum(t1, t2).min(1)

P1 = pr['src_tri'][:, :3].astype(np.float64); P2 = pr['tar_tri'][:, :3].astype(np.float64)
rng = np.random.default_rng(1)
u = rng.standard_normal((2, 40000, 3)); u /= np.linalg.norm(u, axis=2, keepdims=True)
q1, q2 = pr['radius'] * u[0] + pr['center'], pr['radius'] * u[1] + pr['center']
dd = q2 - q1; Lacc = np.concatenate([dd / np.linalg.norm(dd, axis=1, keepdims=True), q1], 1)
Lacc = Lacc[hits_box(Lacc, P1) & hits_box(Lacc, P2)][:3000]
print('lines accepted like the sampler:', len(Lacc), '-- tests per (line, cloud): [level sizes] -> [top, ..., exact]')
for name in ('src_tri', 'tar_tri'):
    tri = pr[name]; P = tri[:, :3].astype(np.float64); thr = thresholds(tri)
    o = order_hilbert(P)
    orders = {'hilbert': o, 'kd64': kd_blocks(P, o, 64), 'kd256': kd_blocks(P, o, 256), 'kdfull': local_kd(P, np.arange(len(P)))}
    for on, oo in orders.items():
        for sizes in ((64, 16, 8), (256, 64, 16, 8), (256, 64, 16, 8, 4)):
            print(name, on, sizes, count_levels(P[oo], thr[oo], Lacc, sizes))
