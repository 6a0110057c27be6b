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


def product_lines(prs, L):
    """(B, L, 6) lines for the synthetic pairs from the PRODUCT sampler (CPU RNG stream seeded by the
    sample index); tools never touch oracle/ (test infrastructure only)."""
    import loss as Lmod
    out = []
    for b, p in enumerate(prs):
        torch.manual_seed(b)
        out.append(Lmod.Random_uniform_distribution_lines_batch_efficient_resample(
            torch.tensor([[float(p["radius"])]]), torch.from_numpy(p["center"]).reshape(1, 3), L,
            torch.from_numpy(p["src"])[None].cuda(), torch.from_numpy(p["tar"])[None].cuda(), "cuda")[0])
    return torch.stack(out)
src = torch.from_numpy(np.stack([p["src_tri"] for p in prs])).cuda()
tar = torch.from_numpy(np.stack([p["tar_tri"] for p in prs])).cuda()
ln = product_lines(prs, L)
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
