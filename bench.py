#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: point-pairs/sec for the intersected-line loss
fwd+bwd at B=8, N=M=4096 (L=10000 lines, the RPM call-site default), per GPU.

One "step" = the fused training op forward + backward, six launches: triangle records (rigid
apply of the source, thresholds, state clearing) -> cell sort + sphere tree -> tree-culled
line<->triangle scan of both clouds (K1) -> per-line distances (K2) -> median + Welsch reduce
(K3+K4) -> direct backward to (dR, dT) with the 14-float shard payload (K5') -> one fused
all-reduce of [loss sum, valid count, sum dR, sum dT] over ranks (asynchronous: it overlaps the
next step's kernels; every reduction completes inside the timed region).  Inputs are resident in HBM before the
timed region; line sampling (K8) and Chamfer (K7) are timed separately and reported as extras.
pairs per step = B * L * 3 * (N + M) per GPU; value = all ranks' pairs / max
time over ranks.  Weak scaling: B=8 per GPU (config 3 of BASELINE.json is B=64 over 8 GPUs).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "a-robust-registration-loss_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FLOPS_PER_PAIR = 18          # 3 sub, 3+3+1 mul, 2+2+1 add, 1 sub, 1 sqrt-class
VALU_PEAK_TFLOPS = 78.6      # 157.3 TFLOP/s fp32 vector peak counts FMA as 2 flops; this path is
                             # contraction-off mul/add (one flop per lane-op) -> half of it
HBM_PEAK_GBS = 8000.0


def make_workload(B, N, M, L, rank, dev):
    import loss as Lmod
    from rrl_hip import synth
    tri1, tri2, src, tar, rad, ctr = [], [], [], [], [], []
    for b in range(B):
        pr = synth.make_pair(1000 * rank + b, N, M)
        tri1.append(pr["src_tri"]); tri2.append(pr["tar_tri"]); src.append(pr["src"])
        tar.append(pr["tar"]); rad.append(pr["radius"]); ctr.append(pr["center"])
    to = lambda a: torch.from_numpy(np.stack(a)).to(dev)  # noqa: E731
    w = dict(tri1=to(tri1), tri2=to(tri2), src=to(src), tar=to(tar))
    torch.manual_seed(1000 * rank)  # CPU seed selects the sampler's uniform streams
    t0 = time.perf_counter()
    w["lines"] = Lmod.Random_uniform_distribution_lines_batch_efficient_resample(
        torch.tensor(rad).reshape(B, 1), torch.from_numpy(np.stack(ctr)), L, w["src"], w["tar"],
        dev)
    torch.cuda.synchronize()
    w["sample_s"] = time.perf_counter() - t0
    # per-sample "predicted" transforms (what RPM/DCP/FMR hand to the loss): small rotations
    gen = torch.Generator().manual_seed(7 + rank)
    from LieAlgebra import se3
    R, T = se3.exp3(0.05 * torch.randn(B, 6, generator=gen))
    w["R"], w["T"] = R.to(dev).requires_grad_(True), T.to(dev).requires_grad_(True)
    return w


def cpu_baseline(N, M, L, budget_s=12.0):
    """The CPU oracle (C, OpenMP over lines, all host cores) on a bounded sample of the same
    workload: whole samples of N=M=4096, L=10000, fwd+bwd, until ~budget_s seconds are spent."""
    from oracle import rrl_oracle
    from rrl_hip import synth
    rrl_oracle.build()
    cores = os.cpu_count() or 1
    pr = synth.make_pair(500, N, M)
    rands = synth.uniform_streams(0, 3, L)
    lines = rrl_oracle.resample_lines(rands, pr["radius"], pr["center"], pr["src"], pr["tar"], L)
    done, spent = 0, 0.0
    while spent < budget_s:
        t0 = time.perf_counter()
        rrl_oracle.loss(pr["src_tri"], pr["tar_tri"], lines, want_grad=True)
        spent += time.perf_counter() - t0
        done += 1
    pairs = done * L * 3 * (N + M)
    return {"value": pairs / spent, "unit": "point-pairs/s", "cores": cores, "kind": "port",
            "sample": f"{done} evaluation(s) of one N=M={N}, L={L} sample, loss fwd+bwd, oracle/rrl_oracle.c "
                      f"with OpenMP on {cores} threads, {spent:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)  # 80 us each: the closing fence costs ~3 us/step at 30
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU")
    ap.add_argument("--points", type=int, default=4096)
    ap.add_argument("--lines", type=int, default=10000)
    ap.add_argument("--mode", default=os.environ.get("RRL_SCAN_MODE", "cull"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    args = ap.parse_args()

    from rrl_hip import dist as rdist, ops
    import loss as Lmod
    rank, world, local = rdist.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B, N, M, L = args.batch, args.points, args.points, args.lines
    w = make_workload(B, N, M, L, rank, dev)
    ones = torch.ones(B, device=dev)

    def local_step():
        # transform + loss forward, backward to (dR, dt), and the 14-float shard payload
        w["R"].grad = w["T"].grad = None
        loss, info, _ = ops.registration_loss(w["tri1"], w["R"], w["T"], w["tri2"], w["lines"],
                                              (1, 1, 5, 5), transpose_r=True, mode=args.mode,
                                              want_payload=True)
        torch.autograd.backward([loss], [ones])  # d(sum of losses): no reduction kernel needed
        return ops.last_state().payload

    from rrl_hip import rccl as rrccl
    reducer = rrccl.make_reducer(dev)
    direct = hasattr(reducer, "allreduce_inline")  # direct RCCL binding available

    graphed, inline = None, False
    if not args.no_graph:
        from rrl_hip.graph import GraphedStep
        if direct and os.environ.get("RRL_AR_INLINE", "1") != "0":
            try:  # the all-reduce as the last node of the captured step (in place on the payload)
                graphed = GraphedStep(lambda: reducer.allreduce_inline(local_step()))
                inline = True
            except Exception as exc:
                print(f"[bench] capture with in-graph all-reduce failed ({type(exc).__name__}: {exc})", file=sys.stderr)
                graphed = None
        if graphed is None:
            try:
                graphed = GraphedStep(local_step)
            except Exception as exc:  # capture unsupported: fall back to eager launches
                print(f"[bench] graph capture failed ({type(exc).__name__}: {exc}); eager", file=sys.stderr)
                graphed = None

    last = [None]

    def step():
        # one 14-float all-reduce per step (N > 1): a node of the captured step when the direct
        # RCCL binding is up; otherwise issued asynchronously so that it overlaps the next step's
        # kernels (waited for before its buffer is reused and at the end)
        if inline:
            last[0] = graphed()
            return last[0]
        payload = graphed() if graphed is not None else local_step()
        reducer.submit(payload)
        return payload

    def fence():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    if graphed is None:
        ops.scan_timing(4)  # HIP events around every 4th scan launch, on the launch stream
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
    payload = last[0] if inline else reducer.finish()  # the last reduction is inside the timed region
    fence()
    dt = time.perf_counter() - t0
    if graphed is not None:
        # events cannot be read back from inside a replayed graph: time the dominant kernel in an
        # eager pass of the same step right after the timed region
        ops.scan_timing(1)
        for i in range(min(args.steps, 20)):
            local_step()
        torch.cuda.synchronize()
    scan_times = ops.scan_timing_collect()
    ops.scan_timing(0)
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if dist.is_initialized():
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    scan_ms = float(np.mean(scan_times))

    # extras, outside the timed region
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        cd = Lmod.chamfer_dist(w["src"], w["tar"])
    torch.cuda.synchronize()
    chamfer_ms = (time.perf_counter() - t1) / 5 * 1e3

    if rank == 0:
        pairs_step = B * L * 3 * (N + M)
        value = world * pairs_step * args.steps / dt
        scan_s = scan_ms * 1e-3
        alg_bytes = B * (N + M) * 48 + B * L * 24 + 2 * B * L * 4  # ptri + lines + counts
        pmc = None
        pmc_file = os.path.join(ROOT, "profiles", "scan_hbm_traffic.json")
        if os.path.exists(pmc_file):
            pmc = json.load(open(pmc_file)).get(f"B{B}_N{N}_L{L}_{args.mode}")
        # executed (not algorithmic) VALU work of the same kernel from the SQ counters of the round
        # profile: wave-instructions x 64 lanes against the same peak, over the live launch time
        executed = None
        sq_file = os.path.join(ROOT, "profiles", "r01s5_pmc_summary.json")
        if os.path.exists(sq_file) and (B, N, L, args.mode) == (8, 4096, 10000, "cull"):
            sq = json.load(open(sq_file)).get("sq_per_kernel", {}).get("cull_scan_kernel", {})
            if "SQ_INSTS_VALU" in sq:
                executed = {"valu_wave_insts": sq["SQ_INSTS_VALU"], "salu_wave_insts": sq.get("SQ_INSTS_SALU"),
                            "lds_wave_insts": sq.get("SQ_INSTS_LDS"),
                            "valu_lane_ops_per_s": sq["SQ_INSTS_VALU"] * 64 / scan_s,
                            "frac_of_valu_peak": sq["SQ_INSTS_VALU"] * 64 / scan_s / (VALU_PEAK_TFLOPS * 1e12),
                            "source": "profiles/r01s5_pmc_summary.json (rocprofv3 --pmc SQ_INSTS_*, own pass)"}
        out = {
            "metric": "point-pairs/sec for loss fwd+bwd at B=8, N=M=4096",
            "value": value, "unit": "point-pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"B={B}/GPU, N=M={N} pseudo-triangles, L={L} lines, fp32 loss "
                                   f"fwd+bwd (BASELINE.json configs[1]); scan mode {args.mode}; "
                                   + ("hipGraph replay" if graphed is not None else "eager launches"),
                       "global_batch": B * world, "parallelism": f"batch-shard dp{world}", "allreduce": type(reducer).__name__ + (" (in-graph)" if inline else "")},
            "roofline": {
                "bound": "valu", "kernel": "K1 line<->triangle scan (cull_scan_kernel)",
                "achieved": FLOPS_PER_PAIR * pairs_step / scan_s / 1e12, "peak": VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": FLOPS_PER_PAIR * pairs_step / scan_s / 1e12 / VALU_PEAK_TFLOPS,
                "launch_ms": scan_ms,
                "note": "fp32 VALU-bound (no FMA allowed: label parity); peak = 157.3/2 TFLOP/s. "
                        "achieved counts the ALGORITHMIC flops of the dense formulation (18 per "
                        "(line, point) pair, SURVEY 8d); the kernel culls exactly, so frac > 1 "
                        "means work skipped, not the VALU beaten (a three-level sphere tree rejects all "
                        "but ~1 % of the pairs): the executed instruction stream keeps the VALU ~35 % "
                        "busy (1.47e7 VALU wave-instructions per launch, profiles/r01s5_pmc_summary.json)",
                "hbm": {"achieved": alg_bytes / scan_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg_bytes / scan_s / 1e9 / HBM_PEAK_GBS,
                        "algorithmic_bytes": alg_bytes},
                # HBM bytes per launch of that kernel from the PMC counters (separate rocprofv3 --pmc
                # passes, gfx950-corrected: profiles/scan_hbm_traffic.json), null when not collected
                "traffic": (pmc or {}).get("bytes"),
                "traffic_detail": pmc,
                "executed": executed,
            },
            "extras": {"loss_sum": float(payload[0]), "valid": float(payload[1]),
                       "chamfer_ms": chamfer_ms, "chamfer_pairs_per_s": B * N * M / (chamfer_ms * 1e-3),
                       "chamfer": float(cd), "line_sampling_s": w["sample_s"],
                       "scan_share_of_step": scan_ms / (dt / args.steps * 1e3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, M, L)
        print(json.dumps(out))
    if hasattr(reducer, "close"):
        reducer.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
