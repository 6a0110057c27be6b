"""In-kernel time stamps of the single-tile step kernel (experimental build: RRL_HIPCC_FLAGS=-DRRL_STAMPS -> lib_exp).
usage: RRL_HIPCC_FLAGS=-DRRL_STAMPS python3 tools/stamps.py B,N,M,L"""
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
rs = Step(src, tar, L, transpose_r=True)
lib = _lib.load()
lib.rrl_debug_stamps.argtypes = [ctypes.c_void_p]
acc = np.zeros(32); n = 0
for it in range(60):
    rs(R, t, ln); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    assert lib.rrl_debug_stamps(buf) == 0
    v = np.array(list(buf), dtype=np.float64)
    if it >= 10: acc += (v - v[0]) / 100.0; n += 1
names = {0: "entry", 1: "pair: phase 1 done (counts, compaction)", 2: "pair: phase 2 done (gathers, D)", 3: "pair stores fenced + barrier",
         4: "reduce: prefix loaded", 9: "reduce: tiles in registers, n known", 10: "reduce: radix pass 0 done", 11: "reduce: median known",
         12: "reduce: Welsch sums done", 6: "reduce: loss stored", 7: "fence + barrier", 8: "backward issued (wave 0)"}
for i in (0, 1, 2, 3, 4, 9, 10, 11, 12, 6, 7, 8):
    print(f"  {acc[i] / n:7.2f} us  {names[i]}")
