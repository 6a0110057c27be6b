"""The one-call demo epoch with every riding launch (Chamfer walk in the scan's, the next epoch's sampler in the per-line and
backward launches: the default) against RRL_DEMO_RIDE=0 (every kernel in a launch of its own), long runs at several sizes,
deterministic backward: the per-epoch loss / Chamfer / validity and the final pose must be bit-identical.
usage (GPU box): python tools/demo_ride_soak.py [epochs]"""
import argparse, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
import importlib
import torch
demo = importlib.import_module("test_demo_optimized_Lie_Algebra")
from rrl_hip import ops
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
ops.set_deterministic(True)
bad = 0
for n_pts, n_lines, diag in ((1024, 20000, 0.0), (1024, 20000, 11.7), (400, 4000, 0.0), (1500, 9000, 0.0), (700, 2500, 0.0)):
    out = {}
    for ride in ("1", "0"):
        os.environ["RRL_DEMO_RIDE"] = ride
        ops._sampler_key.clear()
        with tempfile.TemporaryDirectory() as d:
            args = argparse.Namespace(data_path=None, device="cuda:0", seed=3, label1="s", Save_path=d, n_epoch=epochs,
                                      n_sample_line=n_lines, synthetic=n_pts, graph=True, print_every=0, device_rng=True,
                                      save_every=0, synthetic_diag=diag)
            hist, model = demo.main(args)
        out[ride] = (np.array([[np.nan if v is None else v for v in h[1:3]] for h in hist], np.float64),
                     model.parameters_.detach().cpu().numpy().copy())
    same = np.array_equal(out["1"][0], out["0"][0], equal_nan=True) and np.array_equal(out["1"][1], out["0"][1])
    bad += 0 if same else 1
    print(f"N=M={n_pts} L={n_lines} diag={diag or 'unit'}: {epochs} epochs, valid {int(np.isfinite(out['1'][0][:, 0]).sum())}, "
          f"chamfer {out['1'][0][0, 1]:.5f} -> {out['1'][0][-1, 1]:.5f}: riding == separate launches: {same}")
print("mismatching runs:", bad)
sys.exit(1 if bad else 0)
