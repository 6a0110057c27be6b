"""Start-time / lifetime spread of the culled scan's wavefronts at the bench shape (the instrumented
instantiation writes every wavefront's start and end on the 100 MHz wall clock): a kernel lasts as long as
its slowest wavefront.  usage (GPU box): python tools/scan_tail.py [B N L]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "a-robust-registration-loss_amd")]
import bench
from rrl_hip import ops
B, N, L = ([int(v) for v in sys.argv[1:4]] + [8, 4096, 10000][len(sys.argv) - 1:])[:3]
dev = torch.device("cuda", 0)
w = bench.make_workload(B, N, N, L, 0, dev)
# the prepared k-d order (rrl_cloud_order) unless RRL_PREPARED=0: the per-step cell order
opts = ops.make_opts(order1=ops.cloud_order(w["tri1"]), order2=ops.cloud_order(w["tri2"])) if os.environ.get("RRL_PREPARED", "1") != "0" else None
for _ in range(5):
    st = ops.loss_forward_raw(w["tri1"], w["tri2"], w["lines"], mode="cull", opts=opts)
torch.cuda.synchronize()
ops.scan_counters(True)
st = ops.loss_forward_raw(w["tri1"], w["tri2"], w["lines"], mode="cull", opts=opts)
torch.cuda.synchronize()
raw = ops.scan_counters(False, raw=True).cpu().numpy()
if os.environ.get("RRL_DELAY_REPORT"):  # (experimental build -DCULL_DELAY_HALF: odd workgroups entered ~2 us late; wall0 is taken AFTER the sleep)
    wg = np.arange(raw.shape[0]) // 8
    t0 = raw[raw[:, 5] == 1][:, 8].min()
    for par, name in ((0, "even workgroups (on time)"), (1, "odd workgroups (delayed)")):
        r = raw[(raw[:, 5] == 1) & (wg % 2 == par) & (raw[:, 6] == 0)]
        first = (r[:, 8] - t0) < 600  # entered within 6 us of the launch's first wavefront: the first generation
        t_sl = (r[:, 15] & 0xffffffff) / 100.0
        print(f"  {name}: {len(r)} wavefronts, first generation {int(first.sum())}: start {((r[first, 8] - t0) / 100.0).mean():.2f} us, "
              f"first loads back + slack {t_sl[first].mean():.2f} us after the wavefront's start (later generations {t_sl[~first].mean():.2f}); "
              f"prologue {((r[first, 10] - r[first, 8]) / 100.0).mean():.2f}")
rows = raw[raw[:, 5] == 1]
start = (rows[:, 8] - rows[:, 8].min()) / 100.0
life = (rows[:, 9] - rows[:, 8]) / 100.0
end = start + life
pc = lambda a, q: float(np.percentile(a, q))
print(f"B={B} N=M={N} L={L}: {len(rows)} wavefronts; start (us after the first) 10/50/90/100 %: {pc(start,10):.1f} / {pc(start,50):.1f} / {pc(start,90):.1f} / {start.max():.1f}")
print(f"  lifetime us mean {life.mean():.1f}, 10/50/90/99/100 %: {pc(life,10):.1f} / {pc(life,50):.1f} / {pc(life,90):.1f} / {pc(life,99):.1f} / {life.max():.1f}; last end {end.max():.1f} us")
work = rows[:, 1] * 12 + rows[:, 3] * 11 + rows[:, 4] * 48
print(f"  per-wavefront work (lane-ops below level A): mean {work.mean():.0f}, 90 % {pc(work,90):.0f}, max {work.max():.0f}; corr(lifetime, work) = {np.corrcoef(life, work)[0,1]:.2f}")
late = rows[end > pc(end, 99)]
print(f"  the slowest 1 %: mean exact tests {late[:,3].mean():.0f} (all: {rows[:,3].mean():.0f}), candidates {late[:,4].mean():.0f} (all: {rows[:,4].mean():.0f})")

if rows.shape[1] > 14 and rows[:, 10].max() > 0:
    culled = rows[rows[:, 6] == 0]
    ph = np.stack([culled[:, 10] - culled[:, 8], culled[:, 11] - culled[:, 10], culled[:, 12] - culled[:, 11],
                   culled[:, 14] - culled[:, 13], culled[:, 9] - culled[:, 14]], 1) / 100.0
    names = ["prologue (loads, slack, staging)", "level A (masks + queue fill)", "level B (8 halves per pair, interleaved level D)", "drain D", "candidate flush"]
    first = culled[:, 8] - rows[:, 8].min() < 200  # started in the first 2 us: the full-occupancy generation
    print("  phase means us (all / first generation / later): " + "; ".join(
        f"{n} {ph[:, i].mean():.2f} / {ph[first, i].mean():.2f} / {ph[~first, i].mean() if (~first).any() else float('nan'):.2f}" for i, n in enumerate(names)))
    if rows.shape[1] > 15 and culled[:, 15].max() > 0:  # round 5: inside the prologue
        t_sl, t_pb = (culled[:, 15] & 0xffffffff) / 100.0, (culled[:, 15] >> 32) / 100.0
        t_bar = (culled[:, 10] - culled[:, 8]) / 100.0
        print("  inside the prologue, us after the wavefront's start (all / first generation / later): "
              f"first loads back + slack {t_sl.mean():.2f} / {t_sl[first].mean():.2f} / {t_sl[~first].mean() if (~first).any() else float('nan'):.2f}; "
              f"own share staged {t_pb.mean():.2f} / {t_pb[first].mean():.2f} / {t_pb[~first].mean() if (~first).any() else float('nan'):.2f}; "
              f"past the barrier {t_bar.mean():.2f} / {t_bar[first].mean():.2f} / {t_bar[~first].mean() if (~first).any() else float('nan'):.2f}")
