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
    # (several ranks SHARING one GPU -- validated on a 1-GPU box, tests/test_gpu_harness.py --: also when the caller created the
    #  process group itself)
    if torch.cuda.is_available() and os.environ.get("RRL_SHARE_GPU") == "1":
        local = local % max(torch.cuda.device_count(), 1)
    if (world > 1 or "RANK" in os.environ) and not dist.is_initialized():
        if backend is None:  # RRL_DIST_BACKEND=gloo: e.g. several ranks SHARING one GPU (RCCL refuses duplicate devices) -- the
            backend = os.environ.get("RRL_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")  # N > 1 host logic
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


# ---------------------------------------------------------------------------------------
# Line-sharded single-sample mode (last sentence): ONE pair of clouds too large or too few to shard by
# samples -- its LINES are partitioned over the ranks instead.  A line's scan and per-line stage depend on nothing but
# the two clouds (which every rank holds) and the line; the only coupling is the median of all D values and the bucket
# means.  So every rank runs the scan and the per-line stage on its share, ONE all-gather moves the selected lines'
# rows (16 D values + one byte each: <= 68 bytes per selected line, ~9 % of the lines), every rank reduces the merged
# list (include/rrl.h rrl_loss_reduce_rows: same median, order-independent sums => the loss is bit-identical to the
# unsharded one and identical on all ranks) and the backward of a rank's own lines uses the merged statistics; the
# point gradients are summed by one all-reduce.
# ---------------------------------------------------------------------------------------
def gather_rows(rows, kj, group=None, flag=None):
    """All ranks' (rows (S_r, 16) fp32, kj (S_r,) uint8) concatenated in rank order on every rank: one all-gather of the
    sizes, one of the rows padded to the largest share (kj travels in a 17th column).  flag (int tensor (1,), optional):
    a per-rank flag that rides in the sizes' all-gather (round 4: the scan's NaN flag -- one collective less per call);
    then (rows, kj, max over the ranks' flags) is returned."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return (rows, kj) if flag is None else (rows, kj, flag.reshape(1).clone())
    dev = rows.device
    n = torch.zeros(2, dtype=torch.int64, device=dev)
    n[0] = rows.shape[0]
    if flag is not None:
        n[1] = flag.reshape(-1)[0].to(torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    fmax = torch.stack(sizes)[:, 1].max().reshape(1)
    sizes = [int(s[0]) for s in sizes]
    cap = max(max(sizes), 1)
    mine = torch.zeros(cap, 17, dtype=torch.float32, device=dev)
    mine[:rows.shape[0], :16] = rows
    mine[:rows.shape[0], 16] = kj.to(torch.float32)  # k | j << 4 <= 68: exact in fp32
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    allr = torch.cat([p[:s] for p, s in zip(parts, sizes)])
    out = (allr[:, :16].contiguous(), allr[:, 16].to(torch.uint8).contiguous())
    return out if flag is None else out + (fmax.to(flag.dtype),)


def line_shard_local(tri1, tri2, line, rng=(1, 1, 5, 5), mode="cull", chunk=0):
    """The rank-local part on GPU tensors tri1 (1, N, 9), tri2 (1, M, 9), line (1, L_r, 6) = this rank's share of the lines:
    prepare + sort + scan + per-line stage (no reduce).  Returns (state, rows (S_r, 16), kj (S_r,)) -- the selected lines'
    canonical D tiles and k | j << 4 bytes in the per-line stage's compact order."""
    from . import ops
    lib = ops._lib.load()
    if tri1.shape[0] != 1 or tri2.shape[0] != 1 or line.shape[0] != 1:
        raise ValueError("the line-sharded mode evaluates ONE sample (B == 1); batches shard by samples (sharded_batch_loss)")
    N, M, L = tri1.shape[1], tri2.shape[1], line.shape[1]
    s_m, s_n, e_m, e_n = ops._check_range(rng)
    dev = tri1.device
    st = ops.LossState(1, N, M, max(L, 1), 1, dev)
    st.status.zero_()
    if L == 0:
        return st, torch.zeros(0, 16, device=dev), torch.zeros(0, dtype=torch.uint8, device=dev)
    ws, nb = ops._p(st.ws), st.nbytes
    with ops._guard(dev):
        s = ops._stream(dev)
        ops.check(lib.rrl_tri_prepare(ops._p(tri1), ops._p(tri2), ws, nb, 1, N, M, L, s), "rrl_tri_prepare")
        ops.check(lib.rrl_line_tri_scan(ops._p(line), ws, nb, 1, N, M, L, ops._MODES[mode], int(chunk), s), "rrl_line_tri_scan")
        ops.check(lib.rrl_line_pair_dist(ops._p(tri1), ops._p(tri2), ops._p(line), ws, nb, 1, N, M, L, s_m, s_n, e_m, e_n,
                                         0, s), "rrl_line_pair_dist")
    nblk = (L + 1023) // 1024
    cnt = st.blkcnt[:nblk].to(torch.int64)
    keep = (torch.arange(1024, device=dev)[None, :] < cnt[:, None]).reshape(-1)  # slot = 1024 tile + rank
    rows = st.vals[0][keep].contiguous()
    kj = st.kjc[0][keep].contiguous()
    return st, rows, kj


def line_shard_merge(st, rows, kj, rng=(1, 1, 5, 5), nan_flag=None):
    """Reduce the merged rows INTO the state of this rank (loss, MED, BCNT, BSUM, INFO of its workspace): its backward then
    uses the statistics of ALL lines.  nan_flag: the scan's NaN flag over all ranks (int tensor) or None."""
    from . import ops
    lib = ops._lib.load()
    s_m, s_n, e_m, e_n = ops._check_range(rng)
    dev = st.ws.device
    if nan_flag is not None:
        st.status[0] = nan_flag.to(st.status.dtype)
    n = int(rows.shape[0])
    scratch = torch.empty((n + 1023) // 1024 + 2, dtype=torch.int32, device=dev)
    if n == 0:  # (the kernel reads no row, but wants valid pointers)
        rows, kj = torch.zeros(1, 16, device=dev), torch.zeros(1, dtype=torch.uint8, device=dev)
    with ops._guard(dev):
        ops.check(lib.rrl_loss_reduce_rows(ops._p(rows), ops._p(kj), n, ops._p(scratch), ops._p(st.loss), ops._p(st.med),
                                           ops._p(st.bcnt), ops._p(st.bsum), ops._p(st.info), ops._p(st.status),
                                           s_m, s_n, e_m, e_n, ops._stream(dev)), "rrl_loss_reduce_rows")
    st._merge_keepalive = (rows, kj, scratch)  # until the stream has consumed them
    return st


class _LineShardedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points1, points2, line, rng, mode, group):
        from . import ops
        dev = ops._home(points1, points2, line)
        tri1, tri2 = ops._prep(points1, "points1", 9, dev), ops._prep(points2, "points2", 9, dev)
        ln = ops._prep(line, "line", 6, dev)
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        lo, hi = shard_bounds(ln.shape[1], rank, world)
        mine = ln[:, lo:hi].contiguous()
        st, rows, kj = line_shard_local(tri1, tri2, mine, rng, mode)
        rows, kj, flag = gather_rows(rows, kj, group, flag=st.status[:1])  # (the NaN flag rides with the shares' sizes)
        line_shard_merge(st, rows, kj, rng, flag[0])
        ctx.st, ctx.tri1, ctx.tri2, ctx.group, ctx.world = st, tri1, tri2, group, world
        ctx.nlines = hi - lo
        ctx.in_devs = (points1.device, points2.device)
        ctx.mark_non_differentiable(st.info, st.status)
        ctx.set_materialize_grads(False)
        return st.loss.view(-1), st.info, st.status

    @staticmethod
    def backward(ctx, g_loss, _g1, _g2):
        from . import ops
        st, tri1, tri2 = ctx.st, ctx.tri1, ctx.tri2
        N, M = tri1.shape[1], tri2.shape[1]
        dev = tri1.device
        if g_loss is None and ctx.world == 1:
            return (None,) * 6
        # A rank whose loss is unused in ITS graph (g_loss None) still takes part in the collectives below with zero
        # gradients (round 4, ADVICE r3: returning early here left the other ranks hanging in their all-reduce).  Whether
        # points2 wants a gradient is a property of the call, the same on every rank.
        g1 = torch.zeros_like(tri1)
        g2 = torch.zeros_like(tri2) if ctx.needs_input_grad[1] else None
        if g_loss is not None and ctx.nlines > 0:
            g = g_loss.detach().to(device=dev, dtype=torch.float32).contiguous()
            with ops._guard(dev):
                ops.check(ops._lib.load().rrl_loss_backward(ops._p(tri1), ops._p(tri2), ops._p(st.ws), st.nbytes, ops._p(g),
                                                            ops._p(g1), ops._p(g2), 1, N, M, ctx.nlines, 0, ops._stream(dev)),
                          "rrl_loss_backward")
        if ctx.world > 1:  # every rank holds the gradient of ITS lines: sum
            dist.all_reduce(g1, op=dist.ReduceOp.SUM, group=ctx.group)
            if g2 is not None:
                dist.all_reduce(g2, op=dist.ReduceOp.SUM, group=ctx.group)
        g1 = g1.to(ctx.in_devs[0]) if ctx.needs_input_grad[0] else None
        if g2 is not None:
            g2 = g2.to(ctx.in_devs[1])
        return g1, g2, None, None, None, None


def line_sharded_loss(points1, points2, line, rng=(1, 1, 5, 5), mode="cull", group=None):
    """The loss of ONE sample (points1 (1, N, 9), points2 (1, M, 9), line (1, L, 6): the same on every rank) with its L
    lines partitioned over the ranks of `group` (contiguous shares, shard_bounds).  Returns (loss (1,), info (1, 4),
    status (4,)) like ops.intersection_loss; the loss is bit-identical to the unsharded one and the same on all ranks,
    its gradient w.r.t. points1 / points2 is the full gradient (summed over the ranks) on every rank.  Collectives per
    call: the shares' sizes with the NaN flag (16 bytes), the selected lines' rows (68 bytes each); per backward: the
    point gradients (every rank takes part, with zeros when its own loss is not used)."""
    return _LineShardedLoss.apply(points1, points2, line, tuple(rng), mode, group)
