"""Experiment: the C2 training step (B=8) issued as S independent sub-batches on S HIP streams inside ONE
captured graph (fork / join on events), against the single-stream step.  The loss is per sample (median per
sample, code/loss.py:223-230), so any split of the batch gives the same numbers; the point is that the
launch-bound kernels of one sub-batch (sort: one workgroup per cloud; pair / reduce / backward: a few
workgroups) overlap the VALU-bound scan of another instead of leaving the GPU idle."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
from rrl_hip import ops, synth
from rrl_hip.graph import GraphedStep
from config_sweep import product_lines

def run(B, N, M, L, S, n=300):
    prs = [synth.make_pair(b, N, M) for b in range(B)]
    src = torch.from_numpy(np.stack([p["src_tri"] for p in prs])).cuda()
    tar = torch.from_numpy(np.stack([p["tar_tri"] for p in prs])).cuda()
    ln = product_lines(prs, L)
    bs = B // S
    parts = []
    for i in range(S):
        sl = slice(i * bs, (i + 1) * bs)
        parts.append(dict(src=src[sl].contiguous(), tar=tar[sl].contiguous(), ln=ln[sl].contiguous(),
                          R=torch.eye(3, device="cuda").repeat(bs, 1, 1).requires_grad_(True),
                          t=torch.zeros(bs, 3, device="cuda").requires_grad_(True), ones=torch.ones(bs, device="cuda")))
    streams = [torch.cuda.Stream() for _ in range(S)] if S > 1 else [None]
    def f():
        main = torch.cuda.current_stream()
        outs = []
        for p, s in zip(parts, streams):
            p["R"].grad = p["t"].grad = None
            if s is None:
                loss, info, _ = ops.registration_loss(p["src"], p["R"], p["t"], p["tar"], p["ln"])
                torch.autograd.backward([loss], [p["ones"]])
            else:
                s.wait_stream(main)
                with torch.cuda.stream(s):
                    loss, info, _ = ops.registration_loss(p["src"], p["R"], p["t"], p["tar"], p["ln"])
                    torch.autograd.backward([loss], [p["ones"]])
            outs.append((loss, info))
        for s in streams:
            if s is not None: main.wait_stream(s)
        return outs
    g = GraphedStep(f)
    for _ in range(10): g()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): g()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    loss = torch.cat([o[0] for o in g.out]); gR = torch.cat([p["R"].grad for p in parts])
    print(json.dumps({"B": B, "N": N, "M": M, "L": L, "streams": S, "us_per_step": round(dt * 1e6, 1),
                      "loss_sum": float(loss.sum()), "gR_abs_sum": float(gR.abs().sum())}), flush=True)

if __name__ == "__main__":
    shapes = [(8, 4096, 4096, 10000)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
    for (B, N, M, L) in shapes:
        for S in (1, 2, 4, 8):
            if B % S == 0: run(B, N, M, L, S)
