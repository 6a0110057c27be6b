"""Lines drawn by the product sampler for a set of box configurations (regular, flat, degenerate,
off-centre, tiny radius), saved as .npz.  tests/test_gpu_parity.py runs it twice -- with and without
the slab pre-test (RRL_SAMPLER_PREFILTER=0) -- and requires identical output.
usage: python tools/sampler_check.py out.npz"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
import loss as L  # noqa: E402

rng = np.random.default_rng(7)
out = {}
cases = []
for k in range(24):
    n1, n2 = 200, 180
    a = rng.standard_normal((n1, 3)).astype(np.float32) * rng.uniform(0.05, 2.0, 3).astype(np.float32)
    b = rng.standard_normal((n2, 3)).astype(np.float32) * rng.uniform(0.05, 2.0, 3).astype(np.float32)
    shift = rng.standard_normal(3).astype(np.float32) * (0.0 if k % 3 == 0 else 0.5)
    b = b + shift
    if k % 4 == 1:
        a[:, k % 3] = 0.0                      # perfectly flat cloud through the origin
    if k % 4 == 2:
        b[:, (k + 1) % 3] = np.float32(0.25)   # flat, off-centre
    if k == 7:
        a[:] = a[0]                            # a single point: zero-volume box
    if k % 5 == 3:
        a += 50.0; b += 50.0                   # far from the origin
    diag = float(np.linalg.norm(np.maximum(a.max(0), b.max(0)) - np.minimum(a.min(0), b.min(0))))
    radius = diag * [1.0, 0.5, 0.25, 2.0][k % 4]
    cases.append((a, b, radius, np.concatenate([a, b]).mean(0).astype(np.float32)))
for k, (a, b, radius, ctr) in enumerate(cases):
    torch.manual_seed(100 + k)
    lines = L.Random_uniform_distribution_lines_batch_efficient_resample(
        torch.tensor([[radius]]), torch.from_numpy(ctr).reshape(1, 3), 3000,
        torch.from_numpy(a)[None].cuda(), torch.from_numpy(b)[None].cuda(), "cuda")
    out[f"lines{k}"] = lines[0].cpu().numpy()
np.savez(sys.argv[1], **out)
print("filled rows:", [int((np.abs(v).sum(1) > 0).sum()) for v in out.values()])
