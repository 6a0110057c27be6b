#!/usr/bin/env python3
"""Experiment: the B=8 step as ONE captured chain versus K concurrent chains over sub-batches
(parallel branches of the same hipGraph: the latency-bound small kernels of one branch overlap
the wide scan of another).  Usage (GPU box): python tools/branch_timing.py [B N L]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "a-robust-registration-loss_amd")]
import torch  # noqa: E402
import bench  # noqa: E402
from rrl_hip import ops  # noqa: E402
from rrl_hip.graph import GraphedStep  # noqa: E402

B, N, L = (int(v) for v in (sys.argv[1:4] + ["8", "4096", "10000"][len(sys.argv) - 1:]))
dev = torch.device("cuda", 0)
w = bench.make_workload(B, N, N, L, 0, dev)


def chain(sl, R, T, ones):
    R.grad = T.grad = None
    loss, _, _ = ops.registration_loss(w["tri1"][sl], R, T, w["tri2"][sl], w["lines"][sl], (1, 1, 5, 5),
                                       transpose_r=True, mode="cull", want_payload=True)
    torch.autograd.backward([loss], [ones])
    return ops.last_state().payload


def make_step(k):
    parts = []
    for i in range(k):
        sl = slice(i * B // k, (i + 1) * B // k)
        parts.append((sl, w["R"].detach()[sl].clone().requires_grad_(True),
                      w["T"].detach()[sl].clone().requires_grad_(True),
                      torch.ones(sl.stop - sl.start, device=dev)))
    streams = [torch.cuda.Stream() for _ in range(k - 1)]

    def step():
        main = torch.cuda.current_stream()
        outs = [None] * k
        for i in range(1, k):
            streams[i - 1].wait_stream(main)
            with torch.cuda.stream(streams[i - 1]):
                outs[i] = chain(*parts[i])
        outs[0] = chain(*parts[0])
        for s in streams:
            main.wait_stream(s)
        return outs
    return step


for k in (1, 2, 4):
    if B % k:
        continue
    g = GraphedStep(make_step(k))
    for _ in range(20):
        g()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        outs = g()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    tot = sum(float(o[0]) for o in outs)
    print(f"branches={k}: {dt * 1e6:7.1f} us/step  loss_sum={tot:.6f}")
