"""Epochs per second of the single-pair demo (BASELINE.json configs[0]: N=M=1024, L=20000 lines),
eager drop-in loop vs the captured step, on a synthetic pair."""
import argparse, os, sys, time, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
import importlib
demo = importlib.import_module("test_demo_optimized_Lie_Algebra")
for graph, device_rng, save_every, diag in ((False, False, 10, 0.0), (True, False, 10, 0.0), (True, True, 10, 0.0), (True, True, 0, 0.0),
                                            (False, False, 10, 11.7), (True, True, 0, 11.7)):
    with tempfile.TemporaryDirectory() as d:
        args = argparse.Namespace(data_path=None, device="cuda:0", seed=1, label1="s", Save_path=d, n_epoch=300 if not graph else 3000,
                                  n_sample_line=20000, synthetic=1024, graph=graph, print_every=0,
                                  device_rng=device_rng, save_every=save_every, synthetic_diag=diag)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        hist, model = demo.main(args)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        done = [h for h in hist if h[1] is not None]
        print(f"graph={graph} device_rng={device_rng} save_every={save_every} diag={diag or 'unit scale'}: {len(hist)/dt:.1f} epochs/s ({dt/len(hist)*1e3:.2f} ms/epoch incl. line sampling, "
              f"Sample_neighs and file output every 10 epochs); chamfer {done[0][2]:.5f} -> {done[-1][2]:.5f}")
