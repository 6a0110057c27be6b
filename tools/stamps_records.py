"""Per-workgroup in-kernel time stamps of the prepared records launch (tri_records_sorted_kernel) (experimental build: RRL_HIPCC_FLAGS=-DRRL_STAMPS -> lib_exp).
usage: RRL_HIPCC_FLAGS=-DRRL_STAMPS python3 tools/stamps_records.py B,N,M,L"""
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
Step = ops.LossStep if os.environ.get("RRL_STEP", "loss") == "loss" else ops.RegistrationStep
rs = Step(src, tar, L, transpose_r=True, prepared=os.environ.get("RRL_PREPARED", "1") != "0")
lib = _lib.load()
lib.rrl_debug_wstamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.rrl_debug_rstamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["entry", "clearing stores issued", "order + raw row loaded, record computed", "record / tree stores issued", "stores acknowledged (barrier)", "line-maxima workgroup done", "-"]
if os.environ.get("RRL_PREPARED", "1") == "0":  # the cold build: the LAST launch that stamps is the cell sort (one workgroup per cloud)
    names = ["sort: entry", "records + AABB loaded, histogram cleared", "cells tallied (LDS atomics)", "cells scanned", "records scattered (LDS)", "copied out", "tree built"]
rows = []
for it in range(40):
    for _ in range(3): rs(R, t, ln)
    torch.cuda.synchronize()
    lib.rrl_debug_rstamps(None, 1)
    rs(R, t, ln); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (8 * 2048))()
    assert lib.rrl_debug_rstamps(buf, 0) == 0
    v = np.array(list(buf), dtype=np.float64).reshape(8, 2048)
    t0 = v[0][v[0] > 0].min()
    rows.append([((v[i][v[i] > 0] - t0) / 100.0) for i in range(7)])
for i in range(7):
    mins = np.mean([r[i].min() for r in rows if len(r[i])]); meds = np.mean([np.median(r[i]) for r in rows if len(r[i])])
    maxs = np.mean([r[i].max() for r in rows if len(r[i])]); cnt = np.mean([len(r[i]) for r in rows])
    print(f"  {names[i]:48s} workgroups {cnt:6.1f}  first {mins:6.2f}  median {meds:6.2f}  last {maxs:6.2f} us")
