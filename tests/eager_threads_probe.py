"""torch-eager CPU leg at several thread counts (which count to use on a many-core host)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
from oracle import rrl_oracle, torch_eager
from rrl_hip import synth
pr = synth.make_pair(500, 4096, 4096)
lines = rrl_oracle.resample_lines(synth.uniform_streams(0, 3, 10000), pr["radius"], pr["center"], pr["src"], pr["tar"], 10000)
t2, ln = torch.from_numpy(pr["tar_tri"]), torch.from_numpy(lines)
for th in (4, 8, 16, 32, 64, 128, os.cpu_count()):
    torch.set_num_threads(th)
    for ml in (64, 256):
        t1 = torch.from_numpy(pr["src_tri"]).clone().requires_grad_(True)
        t0 = time.perf_counter()
        v = torch_eager.loss(t1, t2, ln[:256], max_lines=ml)
        if v is not None:
            v.backward()
        print(f"threads {th} max_lines {ml}: {time.perf_counter() - t0:.2f} s for 256 lines", flush=True)
