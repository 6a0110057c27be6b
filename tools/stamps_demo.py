"""Per-workgroup in-kernel time stamps of the demo epoch's per-line launch (pair_count_kernel: the sampler's count pass for the
next epoch + the per-line stage), experimental build: RRL_HIPCC_FLAGS=-DRRL_STAMPS python3 tools/stamps_demo.py"""
import argparse, ctypes, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
import importlib
demo = importlib.import_module("test_demo_optimized_Lie_Algebra")
from rrl_hip import _lib
with tempfile.TemporaryDirectory() as d:
    args = argparse.Namespace(data_path=None, device="cuda:0", seed=1, label1="s", Save_path=d, n_epoch=300, n_sample_line=20000,
                              synthetic=1024, graph=True, print_every=0, device_rng=True, save_every=0, synthetic_diag=0.0)
    demo.main(args)
import torch
torch.cuda.synchronize()
lib = _lib.load()
lib.rrl_debug_pstamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * (8 * 2048))()
assert lib.rrl_debug_pstamps(buf, 0) == 0
v = np.array(list(buf), dtype=np.float64).reshape(8, 2048)
live = v[0] > 0
t0 = v[0][live].min()
ids = np.where(live)[0]
# count workgroups carry stamp 4 == their end and no stamp 5; the per-line workgroups carry stamp 5
cnt = ids[(v[5][ids] == 0)]
pair = ids[(v[5][ids] > 0)]
print(f"last epoch's pair_count_kernel: {len(cnt)} count workgroups, {len(pair)} per-line workgroups (us after the first entry)")
names_c = ["entry", "candidate generated, slab prefilter", "survivors compacted (2 barriers)", "face tests done", "ballot stored"]
for i, nm in enumerate(names_c):
    x = (v[i][cnt] - t0) / 100.0
    print(f"  count    {nm:40s} first {x.min():6.2f}  median {np.median(x):6.2f}  last {x.max():6.2f}")
names_p = ["entry", "counts loaded", "phase 1 done", "phase 2 done", "tallies flushed", "done"]
for i, nm in enumerate(names_p):
    x = (v[i][pair] - t0) / 100.0
    print(f"  per-line {nm:40s} first {x.min():6.2f}  median {np.median(x):6.2f}  last {x.max():6.2f}")
