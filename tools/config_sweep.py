"""Step time (fused forward + backward: hipGraph replay and direct issue, prepared k-d orders; direct issue of the cold
-- sorting -- step beside it) at the BASELINE.json configs that are not
the bench line: C1 demo (B=1, N=M=1024, L=20000), C4 (N=2048, M=1024 cropped), C5 (N=M=16384,
L=512), plus L=4096 / L=20000 at the C2 shape.  With arguments "B,N,M,L" ...
it times those shapes instead."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
from rrl_hip import ops, synth
from rrl_hip.graph import GraphedStep


def product_lines(prs, L, radius_scale=1.0):
    """(B, L, 6) lines for the synthetic pairs from the PRODUCT sampler (CPU RNG stream seeded by the
    sample index); tools never touch oracle/ (test infrastructure only)."""
    import loss as Lmod
    out = []
    for b, p in enumerate(prs):
        torch.manual_seed(b)
        out.append(Lmod.Random_uniform_distribution_lines_batch_efficient_resample(
            torch.tensor([[float(p["radius"]) * radius_scale]]), torch.from_numpy(p["center"]).reshape(1, 3), L,
            torch.from_numpy(p["src"])[None].cuda(), torch.from_numpy(p["tar"])[None].cuda(), "cuda")[0])
    return torch.stack(out)

def run(name, B, N, M, L, crop=False, noise=0.01, diag=None):
    """diag: rescale every pair so that the target's AABB diagonal is `diag` and sample the lines with
    radius = the FULL diagonal -- the reference demo's convention and data scale
    (test_demo_optimized_Lie_Algebra.py:45; 11.7 on sample_data/challenge_data)."""
    prs = [synth.make_pair(b, N, M, crop=crop, noise=noise) for b in range(B)]
    if diag is not None:
        for p in prs:
            sc = np.float32(diag / (2.0 * float(p["radius"])))
            for k in ("src", "tar", "src_tri", "tar_tri", "center"):
                p[k] = (p[k] * sc).astype(np.float32)
            p["radius"] = float(p["radius"]) * float(sc)
    src = torch.from_numpy(np.stack([p["src_tri"] for p in prs])).cuda()
    tar = torch.from_numpy(np.stack([p["tar_tri"] for p in prs])).cuda()
    ln = product_lines(prs, L, 2.0 if diag is not None else 1.0)
    R = torch.eye(3, device="cuda").repeat(B, 1, 1).requires_grad_(True)
    t = torch.zeros(B, 3, device="cuda").requires_grad_(True)
    ones = torch.ones(B, device="cuda")
    o1, o2 = (ops.cloud_order(src), ops.cloud_order(tar)) if max(N, M) <= 65536 else (None, None)
    def f():
        R.grad = t.grad = None
        loss, info, _ = ops.registration_loss(src, R, t, tar, ln, mode=os.environ.get("RRL_SCAN_MODE", "cull"), order1=o1, order2=o2)
        torch.autograd.backward([loss], [ones])
        return loss, info
    g = GraphedStep(f)
    for _ in range(5): g()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 200
    for _ in range(n): g()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    pairs = B * L * 3 * (N + M)
    # the same step issued as two plain C calls (ops.RegistrationStep: no autograd node, no graph)
    Rd, td = R.detach(), t.detach()
    times = {}
    for prepared in (True, False):
        rs = ops.RegistrationStep(src, tar, L, transpose_r=True, mode=os.environ.get("RRL_SCAN_MODE", "cull"), prepared=prepared,
                                  src_order=o1 if prepared else None, tar_order=o2 if prepared else None)
        for _ in range(10): rs(Rd, td, ln)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): rs(Rd, td, ln)
        torch.cuda.synchronize(); times[prepared] = (time.perf_counter() - t0) / n
    dd = times[True]
    # SURVEY 8(d)'s step (bench.py's timed step since round 5): ops.LossStep -> rrl_loss_step_ex, backward to points1.grad
    ltimes = {}
    for prepared in (True, False):
        ls = ops.LossStep(src, tar, L, transpose_r=True, mode=os.environ.get("RRL_SCAN_MODE", "cull"), prepared=prepared,
                          src_order=o1 if prepared else None, tar_order=o2 if prepared else None)
        for _ in range(10): ls(Rd, td, ln)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): ls(Rd, td, ln)
        torch.cuda.synchronize(); ltimes[prepared] = (time.perf_counter() - t0) / n
    print(json.dumps({"mode": os.environ.get("RRL_SCAN_MODE", "cull"), "config": name, "B": B, "N": N, "M": M, "L": L, "us_per_step": round(dt * 1e6, 1),
                      "us_per_step_direct": round(dd * 1e6, 1), "us_per_step_direct_cold": round(times[False] * 1e6, 1),
                      "us_per_loss_step_8d": round(ltimes[True] * 1e6, 1), "us_per_loss_step_8d_cold": round(ltimes[False] * 1e6, 1),
                      "pairs_per_s": pairs / dt, "selected_lines": int(g.out[1][:, 1].sum()),
                      "loss0": float(g.out[0][0]), "filled_lines": int((ln.abs().sum(-1) > 0).sum()),
                      "fallback_wavefronts": int(ops.last_state().status[1])}))

if __name__ == "__main__":
    if len(sys.argv) > 1:  # e.g. tools/config_sweep.py 1,32768,32768,10000 1,65536,65536,10000
        for spec in sys.argv[1:]:
            B, N, M, L = (int(x) for x in spec.split(","))
            run(f"B={B} N={N} M={M} L={L}", B, N, M, L)
        sys.exit(0)
    run("C1 demo", 1, 1024, 1024, 20000)
    run("C1 demo at the reference's data scale (AABB diagonal 11.7, radius = full diagonal)", 1, 1024, 1024, 20000, diag=11.7)
    run("C2 bench", 8, 4096, 4096, 10000)
    run("C2 L=4096", 8, 4096, 4096, 4096)
    run("C2 L=20000", 8, 4096, 4096, 20000)
    run("C4 partial overlap", 8, 2048, 1024, 10000, crop=True, noise=0.02)
    run("C5 fragments", 1, 16384, 16384, 512)
    run("C5 fragments B=8", 8, 16384, 16384, 512)
    run("C3 whole batch on one GPU (B=64)", 64, 4096, 4096, 10000)
    run("B=32", 32, 4096, 4096, 10000)
