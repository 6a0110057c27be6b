import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
from rrl_hip import ops, synth
import loss as Lm
B, N, M = 8, 4096, 4096
prs = [synth.make_pair(b, N, M) for b in range(B)]
x = torch.from_numpy(np.stack([p["src"] for p in prs])).cuda()
y = torch.from_numpy(np.stack([p["tar"] for p in prs])).cuda()
for _ in range(20): Lm.chamfer_dist(x, y)
torch.cuda.synchronize()
ops.chamfer_counters(True)
Lm.chamfer_dist(x, y)
torch.cuda.synchronize()
raw = ops._cham_counter_buf.reshape(-1, 16).cpu().numpy()
ops.chamfer_counters(False)
rows = raw[raw[:, 4] == 1]
start = (rows[:, 9] - rows[:, 9].min()) / 100.0   # us
life = rows[:, 8] / 100.0
end = start + life
print(f"{len(rows)} wavefronts: start times (us after the first) percentiles 10/50/90/100 = "
      f"{np.percentile(start, 10):.1f} / {np.percentile(start, 50):.1f} / {np.percentile(start, 90):.1f} / {start.max():.1f}; "
      f"lifetime mean {life.mean():.1f} us (10/90: {np.percentile(life, 10):.1f} / {np.percentile(life, 90):.1f}); last end {end.max():.1f} us")
print("start time by workgroup index (every 128th):", [round(float(start[i]), 1) for i in range(0, len(rows), len(rows) // 8)])
