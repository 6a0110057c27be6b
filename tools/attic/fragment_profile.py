"""cProfile of the RPM fragment (rrl_hip.callsites.rpm_intersection_loss, 3 poses, B = 8, N = M = 4096) -- host side."""
import cProfile, pstats, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
import torch
sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("ft", os.path.join(ROOT, "tools", "fragment_timing.py"))
src = open(os.path.join(ROOT, "tools", "fragment_timing.py")).read().split("for B, n in")[0]
g = {"__file__": os.path.join(ROOT, "tools", "fragment_timing.py")}
exec(compile(src, "ft", "exec"), g)
C, se3 = g["C"], g["se3"]
C.DEVICE_RNG = True
if os.environ.get("RRL_FRAGMENT_NO_RIDE"):
    C._ride_monitor = lambda *a, **k: False
B, n = 8, 4096
d = g["data_for"](B, n)
gen = torch.Generator().manual_seed(0)
Rs, ts = se3.exp3(0.05 * torch.randn(3 * B, 6, generator=gen))
Rs, ts = Rs.reshape(3, B, 3, 3).cuda().requires_grad_(True), ts.reshape(3, B, 3).cuda().requires_grad_(True)
def rpm():
    Rs.grad = ts.grad = None
    pred = [torch.cat([Rs[i], ts[i][..., None]], -1) for i in range(3)]
    out = C.rpm_intersection_loss(pred, d, n_lines=10000)
    out["loss_intersection"].backward()
for _ in range(10): rpm()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(100): rpm()
torch.cuda.synchronize()
print("rpm fragment: %.3f ms (no profiler)" % ((time.perf_counter() - t0) * 10))
pr = cProfile.Profile(); pr.enable()
for _ in range(100): rpm()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumtime").print_stats(22); print(s.getvalue()[:5000])
