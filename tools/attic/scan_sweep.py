#!/usr/bin/env python3
"""Times the scan kernel alone (HIP events) over lines-per-lane variants, modes and chunk
sizes on the BASELINE config-2 workload.  Tuning aid; prints one JSON line per point."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "a-robust-registration-loss_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from rrl_hip import _lib, ops, synth  # noqa: E402
import loss as L  # noqa: E402


def main():
    B, N, Lines = 8, 4096, 10000
    if len(sys.argv) > 1:
        B, N, Lines = (int(v) for v in sys.argv[1:4])
    dev = torch.device("cuda", 0)
    prs = [synth.make_pair(b, N, N) for b in range(B)]
    to = lambda k: torch.from_numpy(np.stack([p[k] for p in prs])).to(dev)  # noqa: E731
    tri1, tri2, src, tar = to("src_tri"), to("tar_tri"), to("src"), to("tar")
    torch.manual_seed(0)
    lines = L.Random_uniform_distribution_lines_batch_efficient_resample(
        torch.tensor([float(p["radius"]) for p in prs]).reshape(B, 1),
        torch.from_numpy(np.stack([p["center"] for p in prs])), Lines, src, tar, dev)
    lib = _lib.load()
    st = ops.loss_forward_raw(tri1, tri2, lines, mode="strict")
    torch.cuda.synchronize()
    base = (st.count1.sum().item(), st.count2.sum().item(), st.loss.tolist())
    print(json.dumps({"hits1": base[0], "hits2": base[1], "loss": base[2]}))
    pairs = B * Lines * 3 * 2 * N
    s, ws, nb = ops._stream(), ops._p(st.ws), st.nbytes
    variants = [int(v) for v in os.environ.get("SWEEP_VARIANTS", "2,4,8").split(",")]
    chunks = [int(v) for v in os.environ.get("SWEEP_CHUNKS", "64,128,256,512").split(",")]
    ops.scan_timing(1)
    for variant in variants:
        lib.rrl_set_scan_variant(variant)
        for mode in (0, 1, 2):
            for chunk in chunks:
                for it in range(8):
                    assert lib.rrl_tri_prepare(ops._p(tri1), ops._p(tri2), ws, nb, B, N, N, Lines, s) == 0
                    assert lib.rrl_line_tri_scan(ops._p(lines), ws, nb, B, N, N, Lines, mode, chunk, s) == 0
                torch.cuda.synchronize()
                ts = ops.scan_timing_collect()
                ok = (st.count1.sum().item(), st.count2.sum().item()) == base[:2]
                ms = float(np.median(ts[2:]))
                print(json.dumps({"variant": variant, "mode": ["strict", "lazy", "auto"][mode],
                                  "chunk": chunk, "ms": round(ms, 4), "min_ms": round(min(ts), 4),
                                  "Gpairs_s": round(pairs / ms / 1e6, 1),
                                  "TFLOPs": round(18 * pairs / ms / 1e9, 2), "same_counts": ok}))
    ops.scan_timing(0)
    lib.rrl_set_scan_variant(0)


if __name__ == "__main__":
    main()
