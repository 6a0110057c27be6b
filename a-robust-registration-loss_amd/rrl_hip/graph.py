"""HIP-graph capture of a launch-bound step (torch.cuda.CUDAGraph drives hipGraph on ROCm).

A loss step is ~12 short kernels; issued eagerly from Python the host needs ~300 us for them
while the GPU needs ~170 us.  With static shapes the whole forward+backward can be captured
once and replayed with one host call.  The C library only enqueues kernels and memsets on the
stream it is given, so it is capture-safe; the profiling hook (rrl_scan_timing_*) must be off.
"""
import torch


class GraphedStep:
    """Captures `fn()` (forward + backward, no host synchronisation inside) and replays it.
    Inputs must live in fixed tensors that `fn` closes over (update them in place); the
    returned tensors are static outputs overwritten by every replay.
    Do not keep tensors with a grad_fn from earlier EAGER calls of `fn` alive: they pin the
    AccumulateGrad nodes of the leaves to the eager stream, which breaks capture."""

    def __init__(self, fn, warmup=3):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # thread-local capture mode: other threads of the process (the RCCL watchdog of a
        # multi-GPU run) may touch the HIP runtime while this stream is capturing
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.out = fn()

    def __call__(self):
        self.graph.replay()
        return self.out
