"""Batch-shard data parallelism: one process per GPU, torch.distributed over RCCL/xGMI.

Samples are independent (every reference caller loops `for j in range(B)` and sums:
rpm/Train_RPM.py:226-231, dcp/Train_DCP.py:266-270, fmr/model.py:302-306), so the batch is
partitioned over ranks with no data-path exchange; the only collectives are a sum
all-reduce of the scalar loss (and of the gradient of parameters shared across samples,
e.g. the 6-vector of a multi-pair Reconstruction_point).  Payloads are 4-28 bytes:
latency-bound, one fused all-reduce per step.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun).
    Returns (rank, world_size, local_rank).  backend defaults to nccl (= RCCL on ROCm) when a
    GPU is visible, else gloo."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # under torchrun (RANK set) the group is created even for one rank, so that a 1-GPU launch
    # exercises the same RCCL path as N > 1
    if (world > 1 or "RANK" in os.environ) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_bounds(total, rank, world):
    """Contiguous, balanced [lo, hi) slice of `total` samples for `rank` (sizes differ by <= 1)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def reduce_loss(local_loss, local_valid, shared_grads=(), group=None):
    """Global sum of the per-sample losses and the number of valid samples, plus in-place sum
    all-reduce of `shared_grads` (gradients of parameters replicated on every rank), fused into
    ONE all-reduce.  local_loss (b,), local_valid (b,) bool.  Returns (loss_sum, n_valid)."""
    dev = local_loss.device
    parts = [torch.where(local_valid, local_loss, torch.zeros_like(local_loss)).sum().reshape(1),
             local_valid.sum().to(local_loss.dtype).reshape(1)]
    parts += [g.reshape(-1).to(dev, local_loss.dtype) for g in shared_grads]
    buf = torch.cat(parts)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    off = 2
    for g in shared_grads:
        n = g.numel()
        g.copy_(buf[off:off + n].reshape(g.shape).to(g.device, g.dtype))
        off += n
    return buf[0], buf[1]


def reduce_payload(payload, group=None):
    """Sum all-reduce of the 14-float shard payload built by ops.shard_payload (HIP, one launch):
    [loss sum, #valid, sum dR (9), sum dT (3)].  One collective per step."""
    if dist.is_initialized():
        dist.all_reduce(payload, op=dist.ReduceOp.SUM, group=group)
    return payload


class PayloadReducer:
    """The per-step all-reduce of the 14-float shard payload, overlapped with the NEXT step's
    kernels: the payload (a static output of the captured step, overwritten by every replay) is
    copied into a private buffer and reduced asynchronously on RCCL's stream while the compute
    stream goes on; `submit` first waits for the previous reduction.  The reduced values of step i
    are returned by `finish()` (call it before the next `submit` to consume every step's sums; a
    throughput loop only consumes the last) -- one step late, which is all a
    throughput loop (or an optimiser that applies step i's update while step i+1's forward runs)
    needs.  xGMI all-reduce latency (tens of microseconds for 56 bytes) is comparable to the
    step itself, so serialising it would cost a large part of the scaling efficiency."""

    def __init__(self, device, group=None):
        self.buf = torch.zeros(14, dtype=torch.float32, device=device)
        self.out = torch.zeros(14, dtype=torch.float32, device=device)
        self.pending = None
        self.group = group

    def submit(self, payload):
        if not dist.is_initialized():
            self.out = payload
            return
        if self.pending is not None:
            self.pending.wait()          # compute stream waits for the previous reduction
        self.buf.copy_(payload)
        self.pending = dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Reduced payload of the last submitted step."""
        if self.pending is not None:
            self.pending.wait()
            self.out.copy_(self.buf)
            self.pending = None
        return self.out


def sharded_batch_loss(points1, points2, line, rng=(1, 1, 5, 5), loss_fn=None, group=None):
    """points1/points2/line hold the GLOBAL batch on every rank (or identical seeds); each rank
    evaluates its shard and the result is the global (loss_sum, n_valid).  loss_fn(p1, p2, ln,
    rng) -> (loss (b,), valid (b,)); defaults to the HIP batched loss."""
    if loss_fn is None:
        import loss as _loss  # the drop-in module one directory up (on sys.path)
        loss_fn = lambda a, b, c, r: _loss.batched_intersection_loss(a, b, c, r)  # noqa: E731
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_bounds(points1.shape[0], rank, world)
    if hi > lo:
        l, v = loss_fn(points1[lo:hi], points2[lo:hi], line[lo:hi], rng)
    else:
        l = torch.zeros(0, device=points1.device)
        v = torch.zeros(0, dtype=torch.bool, device=points1.device)
    return reduce_loss(l, v, group=group)
