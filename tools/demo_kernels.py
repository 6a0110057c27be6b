"""The one-call demo epoch (rrl_demo_epoch) alone, for rocprofv3 --kernel-trace --stats: 3000 epochs of configs[0]
(N = M = 1024, 20000 lines) on a synthetic pair.  Run on the GPU box: tools/demo_kt.sh"""
import argparse, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
import importlib
demo = importlib.import_module("test_demo_optimized_Lie_Algebra")
diag = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
with tempfile.TemporaryDirectory() as d:
    args = argparse.Namespace(data_path=None, device="cuda:0", seed=1, label1="s", Save_path=d, n_epoch=3000, n_sample_line=20000,
                              synthetic=1024, graph=True, print_every=0, device_rng=True, save_every=0, synthetic_diag=diag)
    demo.main(args)
