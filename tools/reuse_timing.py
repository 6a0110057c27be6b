"""Times the fused forward+backward at the bench workload with and without the target-scan
carry-over (rrl_registration_forward_cached), eagerly and as a captured graph."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
from rrl_hip import ops, synth
from rrl_hip.graph import GraphedStep
sys.path.insert(0, ROOT)

B, N, L = 8, 4096, 10000
prs = [synth.make_pair(b, N, N) for b in range(B)]
from oracle import rrl_oracle as o
o.build()
lines = np.stack([o.resample_lines(synth.uniform_streams(b, 10, L), p["radius"], p["center"], p["src"], p["tar"], L)
                  for b, p in enumerate(prs)])
src = torch.from_numpy(np.stack([p["src_tri"] for p in prs])).cuda()
tar = torch.from_numpy(np.stack([p["tar_tri"] for p in prs])).cuda()
ln = torch.from_numpy(lines).cuda()
R = torch.eye(3, device="cuda").repeat(B, 1, 1).requires_grad_(True)
t = torch.zeros(B, 3, device="cuda").requires_grad_(True)
ones = torch.ones(B, device="cuda")
first = ops.loss_forward_raw(src, tar, ln)

def step(tf):
    def f():
        R.grad = t.grad = None
        loss, _, _ = ops.registration_loss(src, R, t, tar, ln, target_from=tf)
        torch.autograd.backward([loss], [ones])
        return loss
    return f

for name, tf in (("full", None), ("target reused", first)):
    g = GraphedStep(step(tf))
    for _ in range(5): g()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): g()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print(f"{name}: {dt*1e6:.1f} us/step (graph)  loss0={float(g.out[0]):.6f}")
