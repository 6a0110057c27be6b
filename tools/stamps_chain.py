"""Per-workgroup in-kernel time stamps of the chained step's build + scan launch (cull_scan_build_kernel) (experimental build:
RRL_HIPCC_FLAGS=-DRRL_STAMPS -> lib_exp).  usage: RRL_HIPCC_FLAGS=-DRRL_STAMPS python3 tools/stamps_chain.py B,N,M,L"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
from rrl_hip import ops, synth, _lib
import loss as Lmod
B, N, M, L = (int(v) for v in sys.argv[1].split(","))
prs = [synth.make_pair(b, N, M) for b in range(B)]
src = torch.from_numpy(np.stack([p["src_tri"] for p in prs])).cuda()
tar = torch.from_numpy(np.stack([p["tar_tri"] for p in prs])).cuda()
ln = []
for b, p in enumerate(prs):
    torch.manual_seed(b)
    ln.append(Lmod.Random_uniform_distribution_lines_batch_efficient_resample(
        torch.tensor([[float(p["radius"])]]), torch.from_numpy(p["center"]).reshape(1, 3), L,
        torch.from_numpy(p["src"])[None].cuda(), torch.from_numpy(p["tar"])[None].cuda(), "cuda")[0])
ln = torch.stack(ln)
R = torch.eye(3, device="cuda").repeat(B, 1, 1); t = torch.zeros(B, 3, device="cuda")
rs = ops.LossStep(src, tar, L, transpose_r=True)
lib = _lib.load()
lib.rrl_debug_rstamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
npad = (N + 63) // 64 * 64
tiles = ((L + 127) // 128 + 7) // 8
nrec = ((npad + 511) // 512 + tiles - 1) // tiles * tiles * B  # workgroups of the records planes (the surplus exits at once)
names = ["entry", "rec: records computed, stores issued", "rec: drained + barrier", "rec: published (release + ticket)", "rec: clearing issued",
         "src: ready seen + acquire", "scan: workgroup's first wavefronts done", "scan: slice staged (barrier)"]
acc = {}
for it in range(30):
    for _ in range(3): rs(R, t, ln)
    torch.cuda.synchronize()
    lib.rrl_debug_rstamps(None, 1)
    rs(R, t, ln); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (8 * 2048))()
    assert lib.rrl_debug_rstamps(buf, 0) == 0
    v = np.array(list(buf), dtype=np.float64).reshape(8, 2048)
    live = v[0] > 0
    nwg = int(live.sum())
    t0 = v[0][live].min()
    ntar = (nwg - nrec) // 2 if N == M else None
    cats = {"rec": slice(0, nrec), "tar": slice(nrec, nrec + ntar), "src": slice(nrec + ntar, nwg)} if ntar else {"all": slice(0, nwg)}
    for cname, sl in cats.items():
        for i in range(8):
            x = v[i][sl]; x = x[x > 0]
            if len(x):
                acc.setdefault((cname, i), []).append(((x - t0) / 100.0))
print(f"B={B} N={N} M={M} L={L}: {nwg} workgroups ({nrec} records)")
for (cname, i), rows in sorted(acc.items()):
    print(f"  {cname:4s} {names[i]:44s} n {np.mean([len(r) for r in rows]):6.1f}  first {np.mean([r.min() for r in rows]):6.2f}  "
          f"median {np.mean([np.median(r) for r in rows]):6.2f}  p90 {np.mean([np.percentile(r, 90) for r in rows]):6.2f}  last {np.mean([r.max() for r in rows]):6.2f} us")
