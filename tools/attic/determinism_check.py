"""The C2 forward replayed N times: per-sample losses, hit counts and hit lists must be bit-identical
every time (the culled scan's queues and atomics only reorder work, never results).
usage (GPU box): python tools/determinism_check.py [N]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "a-robust-registration-loss_amd")]
import bench  # noqa: E402
from rrl_hip import ops  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
w = bench.make_workload(8, 4096, 4096, 10000, 0, dev)
ref = None
bad = 0
for it in range(n):
    with torch.no_grad():
        loss, info, _ = ops.registration_loss(w["tri1"], w["R"].detach(), w["T"].detach(), w["tri2"], w["lines"],
                                              (1, 1, 5, 5), transpose_r=True, mode="cull")
    st = ops.last_state()
    # hit slots beyond the count are stale memory: compare the sorted first `count` entries only
    # lines with more than 4 hits keep whichever 4 arrived first (they are in no bucket): skip them
    c1 = torch.where(st.count1 <= 4, st.count1, torch.zeros_like(st.count1))
    mask = torch.arange(4, device=dev)[None, None, :] < c1[..., None]
    h1 = torch.where(mask, st.hit1, torch.full_like(st.hit1, 1 << 30)).sort(dim=-1)[0]
    cur = (loss.clone(), st.count1.clone(), st.count2.clone(), h1, st.med.clone())
    if ref is None:
        ref = cur
    elif not all(torch.equal(a, b) for a, b in zip(ref, cur)):
        bad += 1
        if bad == 1:
            print("first difference in:", [n_ for n_, a, b in zip(("loss", "count1", "count2", "hit1", "med"), ref, cur) if not torch.equal(a, b)])
torch.cuda.synchronize()
print(f"{n} forwards, {bad} differing from the first; loss sum {float(ref[0].sum()):.6f}")
sys.exit(1 if bad else 0)
