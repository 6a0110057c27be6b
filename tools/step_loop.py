"""The fused step (ops.RegistrationStep: forward + direct backward) in a plain loop at one shape -- the
workload tools/kt.sh runs under rocprofv3 to get per-kernel averages.  usage: step_loop.py B,N,M,L [steps] [diag]
Environment knobs of the library apply (RRL_REDUCE, RRL_SORT_PARTS, ...; RRL_CULL_GEOM / RRL_XCD_ALIGN / RRL_TAIL_MAX_WG only in an
experimental build: RRL_HIPCC_FLAGS=-DRRL_EXPERIMENT); RRL_PREPARED=0: the cold
(sorting) step instead of the prepared build."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
from rrl_hip import ops, synth
import loss as Lmod

B, N, M, L = (int(v) for v in sys.argv[1].split(","))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
diag = float(sys.argv[3]) if len(sys.argv) > 3 and float(sys.argv[3]) > 0 else None
prs = [synth.make_pair(b, N, M) for b in range(B)]
scale = 1.0
if diag is not None:
    for p in prs:
        sc = np.float32(diag / (2.0 * float(p["radius"])))
        for k in ("src", "tar", "src_tri", "tar_tri", "center"):
            p[k] = (p[k] * sc).astype(np.float32)
        p["radius"] = float(p["radius"]) * float(sc)
src = torch.from_numpy(np.stack([p["src_tri"] for p in prs])).cuda()
tar = torch.from_numpy(np.stack([p["tar_tri"] for p in prs])).cuda()
ln = []
for b, p in enumerate(prs):
    torch.manual_seed(b)
    ln.append(Lmod.Random_uniform_distribution_lines_batch_efficient_resample(
        torch.tensor([[float(p["radius"]) * (2.0 if diag is not None else 1.0)]]), torch.from_numpy(p["center"]).reshape(1, 3), L,
        torch.from_numpy(p["src"])[None].cuda(), torch.from_numpy(p["tar"])[None].cuda(), "cuda")[0])
ln = torch.stack(ln)
R = torch.eye(3, device="cuda").repeat(B, 1, 1)
t = torch.zeros(B, 3, device="cuda")
ops.RegistrationStep.ONE_CALL = os.environ.get("RRL_ONE_CALL", "1") != "0"
# RRL_STEP=loss: SURVEY 8(d)'s step (ops.LossStep: backward to points1.grad) -- bench.py's timed step since round 5
Step = ops.LossStep if os.environ.get("RRL_STEP", "reg") == "loss" else ops.RegistrationStep
kw = {}
if os.environ.get("RRL_PRESORT"):  # experiment: the source's rows stored in their sorted order (order = identity)
    o1 = ops.cloud_order(src)
    npad = o1.shape[1]
    src = torch.stack([src[b][o1[b, :N].long()] for b in range(B)]).contiguous()
    ident = torch.arange(npad, dtype=torch.int32, device="cuda").repeat(B, 1).contiguous()
    ident[:, N:] = 0
    kw["src_order"] = ident
rs = Step(src, tar, L, transpose_r=True, mode=os.environ.get("RRL_SCAN_MODE", "cull"),
          prepared=os.environ.get("RRL_PREPARED", "1") != "0", **kw)
# RRL_LINE_SETS=K: K different line sets (and poses) rotated per step -- the demo's pattern: new lines every epoch
K = int(os.environ.get("RRL_LINE_SETS", "1"))
sets = [(R, t, ln)]
for k in range(1, K):
    lk = []
    for b, p in enumerate(prs):
        torch.manual_seed(1000 * k + b)
        lk.append(Lmod.Random_uniform_distribution_lines_batch_efficient_resample(
            torch.tensor([[float(p["radius"]) * (2.0 if diag is not None else 1.0)]]), torch.from_numpy(p["center"]).reshape(1, 3), L,
            torch.from_numpy(p["src"])[None].cuda(), torch.from_numpy(p["tar"])[None].cuda(), "cuda")[0])
    a = 0.01 * k
    Rk = torch.tensor([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32, device="cuda").repeat(B, 1, 1)
    sets.append((Rk, t + 0.001 * k, torch.stack(lk)))
for i in range(10):
    rs(*sets[i % K])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps):
    rs(*sets[i % K])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"shape {B},{N},{M},{L} diag {diag}: {dt * 1e6:.1f} us per step (direct issue), loss_sum {float(rs.st.loss.sum()):.10f} "
      f"nsel {rs.st.info[:, 1].tolist()} status {rs.st.status.tolist()}")
