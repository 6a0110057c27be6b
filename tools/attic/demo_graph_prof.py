"""The captured demo step (configs[0]) for a kernel-trace profile: 500 replays with GPU-drawn lines.
usage (GPU box): rocprofv3 --kernel-trace --stats ... -- python3 tools/demo_graph_prof.py"""
import argparse, importlib, os, sys, tempfile, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
demo = importlib.import_module("test_demo_optimized_Lie_Algebra")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
with tempfile.TemporaryDirectory() as d:
    args = argparse.Namespace(data_path=None, device="cuda:0", seed=1, label1="s", Save_path=d, n_epoch=n,
                              n_sample_line=20000, synthetic=1024, graph=True, print_every=0,
                              device_rng=True, save_every=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hist, model = demo.main(args)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    done = [h for h in hist if h[1] is not None]
    print(f"{n / dt:.1f} epochs/s incl. setup; chamfer {done[0][2]:.5f} -> {done[-1][2]:.5f}")
