"""What a graph node costs: a captured chain of n tiny dependent kernels (x += 1 on one float), replayed back to
back; prints us per replay and the slope per node.  (Interpreting the step: it has 6 kernels + the reducer node.)"""
import time, torch
x = torch.zeros(1, device="cuda")
res = {}
for n in (1, 2, 4, 8, 16, 32):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            for _ in range(n): x.add_(1.0)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): x.add_(1.0)
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500): g.replay()
    torch.cuda.synchronize(); res[n] = (time.perf_counter() - t0) / 500 * 1e6
    print(f"{n:3d} nodes: {res[n]:7.2f} us per replay", flush=True)
print(f"slope 8 -> 32 nodes: {(res[32] - res[8]) / 24:.2f} us per node; intercept (per replay): {res[8] - 8 * (res[32] - res[8]) / 24:.2f} us")
