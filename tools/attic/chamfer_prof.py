"""Chamfer at a BASELINE shape, N calls (run under rocprofv3 --kernel-trace --stats):
    python tools/chamfer_prof.py [B N M calls tree(0|1)]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
from rrl_hip import ops, synth
import loss as Lm
a = [int(v) for v in sys.argv[1:]] + [8, 4096, 4096, 50, 1][len(sys.argv) - 1:]
B, N, M, calls, tree = a
ops.CHAMFER_TREE = bool(tree)
prs = [synth.make_pair(b, N, M) for b in range(B)]
x = torch.from_numpy(np.stack([p["src"] for p in prs])).cuda()
y = torch.from_numpy(np.stack([p["tar"] for p in prs])).cuda()
for _ in range(3):
    v = Lm.chamfer_dist(x, y)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(calls):
    v = Lm.chamfer_dist(x, y)
ev1.record()
torch.cuda.synchronize()
if tree:
    ops.chamfer_counters(True)
    Lm.chamfer_dist(x, y)
    torch.cuda.synchronize()
    c = ops.chamfer_counters(False).cpu().numpy()
    print("  clocks per wave: seed search %.0f | seed eval %.0f | exchange %.0f | walk %.0f (staging %.0f, tests %.0f, passes %.0f) | final barrier %.0f"
          % (c[10] / c[4], c[11] / c[4], c[12] / c[4], c[13] / c[4], c[5] / c[4], c[6] / c[4], c[7] / c[4], c[14] / c[4]))
    print("  wavefront lifetime: %.0f shader clocks = %.2f us by the 100 MHz wall clock -> shader clock %.2f GHz" % (c[15] / c[4], c[8] / c[4] / 100.0, (c[15] / c[4]) / (c[8] / c[4] * 10.0)))
    print(f"  counters: patch-level tests/wave {c[0]/c[4]:.1f}, leaf tests/wave {c[1]/c[4]:.1f}, leaves evaluated/wave {c[2]/c[4]:.1f}, "
          f"pairs {c[3]:.3g} = {c[3]/(B*N*M*2):.4f} of dense, waves {c[4]}")
    # aligned clouds (target = source + small noise): what a converged registration looks like
    y2 = x[:, :M] + 0.003 * torch.randn_like(x[:, :M]) if (M <= N and not os.environ.get('CHAM_NO_ALIGNED')) else None
    if y2 is not None:
        ops.chamfer_counters(True)
        Lm.chamfer_dist(x, y2)
        torch.cuda.synchronize()
        c = ops.chamfer_counters(False).cpu().numpy()
        ev0.record()
        for _ in range(calls):
            Lm.chamfer_dist(x, y2)
        ev1.record()
        torch.cuda.synchronize()
        print(f"  aligned clouds: {ev0.elapsed_time(ev1) / calls * 1e3:.1f} us per call; leaves evaluated/wave {c[2]/c[4]:.1f}, pairs {c[3]/(B*N*M*2):.4f} of dense")
from rrl_hip.graph import GraphedStep
g = GraphedStep(lambda: Lm.chamfer_dist(x, y))
g(); torch.cuda.synchronize()
ev0.record()
for _ in range(calls):
    v = g()
ev1.record()
torch.cuda.synchronize()
print(f"B={B} N={N} M={M} tree={tree}: hipGraph replay {ev0.elapsed_time(ev1) / calls * 1e3:.1f} us per call")
ev0.record()
for _ in range(calls):
    v = Lm.chamfer_dist(x, y)
ev1.record()
torch.cuda.synchronize()
print(f"B={B} N={N} M={M} tree={tree}: {ev0.elapsed_time(ev1) / calls * 1e3:.1f} us per call (eager, incl. launch gaps), value {v.item():.9f}")
