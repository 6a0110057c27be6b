"""Experiment (round 4): the C2 step as S sub-batches on S HIP streams, DIRECT issue (no graph; rounds 2-3 measured this
only inside captured graphs, where every fork / join edge costs ~10 us): does overlapping the latency-bound kernels of
one sub-batch with the scan of another pay?  usage: stream_split_direct.py [B,N,M,L]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "a-robust-registration-loss_amd")]
import bench
from rrl_hip import ops

B, N, M, L = (int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (8, 4096, 4096, 10000)
dev = torch.device("cuda", 0)
w = bench.make_workload(B, N, M, L, 0, dev)
R, T = w["R"].detach().contiguous(), w["T"].detach().contiguous()


def run(S, join, n=400):
    streams = [torch.cuda.Stream() for _ in range(S)] if S > 1 else [torch.cuda.current_stream()]
    sl = [slice(i * B // S, (i + 1) * B // S) for i in range(S)]
    steps = [ops.RegistrationStep(w["tri1"][s].contiguous(), w["tri2"][s].contiguous(), L, want_payload=True) for s in sl]
    args = [(R[s].contiguous(), T[s].contiguous(), w["lines"][s].contiguous()) for s in sl]
    main = torch.cuda.current_stream()
    ev_fork = torch.cuda.Event()
    ev_join = [torch.cuda.Event() for _ in range(S)]

    def one():
        if S == 1:
            return steps[0](*args[0])[0].sum()
        if join:
            ev_fork.record(main)
        for st, step, a, ev in zip(streams, steps, args, ev_join):
            with torch.cuda.stream(st):
                if join:
                    st.wait_event(ev_fork)
                step(*a)
                if join:
                    ev.record(st)
        if join:
            for ev in ev_join:
                main.wait_event(ev)
    for _ in range(20):
        one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        one()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    tot = sum(float(s.st.loss.sum()) for s in steps)
    return dt * 1e6, tot


for S, join in ((1, False), (2, False), (2, True), (4, False), (4, True), (8, True)):
    us, tot = run(S, join)
    print(f"S={S} streams, per-step fork/join events: {join}: {us:.1f} us per step of B={B}, loss_sum {tot:.6f}")
