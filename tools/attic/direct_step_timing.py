"""The C2 training step issued as TWO plain C calls per step (rrl_registration_forward_cached +
rrl_registration_backward through ctypes: no autograd, no graph) against the hipGraph replay of the same
step: does the graph's fixed cost per replay (tools/graph_node_cost.py: ~8 us + 1.5 us per node) show?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "a-robust-registration-loss_amd")]
import torch
import bench
from rrl_hip import ops, _lib
from rrl_hip.graph import GraphedStep
dev = torch.device("cuda", 0)
B, N, M, L = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (8, 4096, 4096, 10000)
w = bench.make_workload(B, N, M, L, 0, dev)
lib = _lib.load()
st = ops.LossState(B, N, M, L, B, dev)
src, tar, ln = w["tri1"], w["tri2"], w["lines"]
R, T = w["R"].detach().contiguous(), w["T"].detach().contiguous()
g = torch.ones(B, device=dev)
out = torch.zeros(B * 12 + 14, device=dev)
gR, gt = out[:B * 9], out[B * 9:B * 12]
def step():
    s = ops._stream(dev)
    rc = lib.rrl_registration_forward_cached(ops._p(src), ops._p(R), ops._p(T), ops._p(tar), ops._p(ln), ops._p(st.ws), st.nbytes,
                                             ops._p(st.loss), B, N, M, L, 1, 1, 1, 5, 5, 3, 0, None, s)
    assert rc == 0
    rc = lib.rrl_registration_backward(ops._p(src), ops._p(R), ops._p(tar), ops._p(st.ws), st.nbytes, ops._p(st.loss), ops._p(g),
                                       None, ops._p(st.gacc[:B * 9]), ops._p(st.gacc[B * 9:B * 12]), None, B, N, M, L, 1, s)
    assert rc == 0
for _ in range(20): step()
torch.cuda.synchronize()
for n in (300, 1000):
    t0 = time.perf_counter()
    for _ in range(n): step()
    t_issue = (time.perf_counter() - t0) / n
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / n
    print(f"direct C calls: host issue {t_issue * 1e6:.1f} us/step, wall {t_all * 1e6:.1f} us/step (n={n}); loss sum {float(st.loss.sum()):.6f}")
gs = GraphedStep(step)
for _ in range(20): gs()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(1000): gs()
torch.cuda.synchronize()
print(f"hipGraph replay of the same two calls: {(time.perf_counter() - t0) / 1000 * 1e6:.1f} us/step")
