"""Drop-in replacement for the reference's `code/loss.py`, backed by hand-written HIP kernels
for MI355X (librrl_hip.so, C ABI in include/rrl.h).

Put this directory on sys.path (the reference's callers do `sys.path.append('../../')`
followed by `from loss import ...`: test_demo_optimized_Lie_Algebra.py:7-10,
rpm/Train_RPM.py:25-31, dcp/Train_DCP.py:17-30, fmr/model.py:18-20) and the same names resolve
here with the same signatures, argument meaning and results.

There is no CPU / eager-PyTorch fallback: all compute runs in the HIP library, and a
missing library or GPU raises `rrl_hip.RRLError`.

Deliberate deviations from the reference:
  * errors are exceptions, not `print(...); exit(0)` (code/loss.py:69-71, 89-91, 266-268);
  * "no populated bucket" returns None, not the 3-tuple (None, None, None) of
    code/loss.py:231-232 -- every caller tests `is not None` or adds the result;
  * `torch.pi` is not monkey-patched (code/loss.py:9); the fp32 value is `PI32` here;
  * hyper-parameters that the reference hard-codes are keyword-only arguments whose
    defaults are the reference's values.
"""
import numpy as np
import torch
import torch.nn as nn

from LieAlgebra import *  # noqa: F401,F403  (se3, so3, sinc -- as code/loss.py:6)
from LieAlgebra import se3
import utils as utils  # noqa: F401
from rrl_hip import ops as _ops
from rrl_hip.ops import RRLError  # noqa: F401

PI32 = 3.1415927410125732  # float32 pi as a Python float (what code/loss.py:9 stores in torch.pi)


def Welsch1(x, c):
    """1 - exp(-(x / c) / 2)   (code/loss.py:20-21)"""
    return 1 - torch.exp(-((x / c)) / 2.0)


def compute_sqrdis_map_2(points_x, points_y):
    """(B, M, 3), (B, N, 3) -> (B, M, N) squared distances (code/loss.py:38-52).  A dense
    output by contract, so this stays a torch expression; the loss and chamfer_dist never
    materialise it."""
    return ((points_x.unsqueeze(2) - points_y.unsqueeze(1)) ** 2).sum(-1)


def cal_intersection_batch2_points_with_line(point_neis, line):
    """The dense scan of one cloud (code/loss.py:68-112): returns
    (points (B*L, N, 9) -- a stride-0 expanded view of point_neis that keeps its grad --,
     norm_d (B*L, N, 3) = d_k / sum_k d_k, detached,  label (B, L, N) bool = all d_k < thr).
    The loss itself never builds these tables (it keeps <= 4 hits per line); this is the public
    function for callers that want them.  Raises ValueError where the reference prints and
    exits (bad rank, NaN distance)."""
    if point_neis.dim() != 3 or line.dim() != 3:
        raise ValueError("Input is wrong")  # code/loss.py:69-71
    B, nf, c = point_neis.shape
    nl = line.shape[1]
    if c != 9:
        raise ValueError("point_neis must hold 3 neighbours per row: (B, N, 9)")
    norm_d, label, status = _ops.dense_scan(point_neis, line)
    if int(status[0]):
        raise ValueError("NaN point-to-line distance (reference: 'Exit the systerm', code/loss.py:88-91)")
    points = point_neis.unsqueeze(1).expand(-1, nl, -1, -1).reshape(-1, nf, 9)  # B = 1: still a view
    return points, norm_d.to(point_neis.device), label.to(point_neis.device)


def cal_loss_intersection_batch_whole_median_pts_lines(s_m, s_n, e_m, e_n, points1, points2, line,
                                                       device='cpu', *, mode=None, chunk=0):
    """The intersected-line robust registration loss (code/loss.py:170-232).

    points1 (B, N, 9) transformed source pseudo-triangles (receives the gradient),
    points2 (B, M, 9) target pseudo-triangles, line (B, L, 6) = [dir, x0].
    Returns a float32 tensor of shape (1,) on `device` with a grad_fn, or None when no
    (k, j) bucket is populated.  Every reference caller passes B = 1 and (1, 1, 5, 5); for
    B > 1 the reference pools all samples' lines and normalises with the LAST sample's
    median and so does this function -- use `batched_intersection_loss` for B
    independent losses in one launch.

    Raises ValueError on a NaN distance (non-unit line direction), where the reference prints
    "Exit the systerm" and exits with status 0 (code/loss.py:89-91).
    mode: "cull" (default: strict semantics; sphere-culled, lazy evaluation where a NaN is
    provably impossible) | "auto" | "strict" | "lazy" (see include/rrl.h), or env RRL_SCAN_MODE.
    """
    if points1.dim() != 3 or line.dim() != 3 or points2.dim() != 3:
        raise ValueError("Input is wrong")  # code/loss.py:69-71
    pool = points1.shape[0] > 1
    # forward + the call's single host sync in one C call: (nbuckets, nselected, nvalues, NaN flag) arrive through a
    # 16-byte pinned copy; the workspace is leased from a per-shape pool (rrl_hip.ops._DropinLoss)
    loss, flags = _ops.intersection_loss_dropin(points1, points2, line, (s_m, s_n, e_m, e_n),
                                                pool=pool, mode=_scan_mode(mode), chunk=chunk)
    if flags[3]:
        raise ValueError("NaN point-to-line distance: line[..., :3] must be unit length or "
                         "all zero (reference: 'Exit the systerm', code/loss.py:88-91)")
    if flags[0] == 0:
        return None
    if loss.device.type != device and loss.device != device:  # (a 'cuda' / device-object argument: already there)
        loss = loss.to(device)
    return loss  # shape (1,): one group (B = 1, or the pooled B > 1 of SURVEY Q2)


def batched_intersection_loss(points1, points2, line, rng=(1, 1, 5, 5), *, mode=None, chunk=0):
    """B independent losses in one set of launches -- what the reference's callers compute
    with `for j in range(B): loss += cal_loss_...(…[j:j+1]…)` (rpm/Train_RPM.py:226-231,
    dcp/Train_DCP.py:266-270, fmr/model.py:302-306).  Returns (loss (B,), valid (B,) bool) on
    the GPU without any host synchronisation; loss[b] is 0 where valid[b] is False."""
    loss, info, _ = _ops.intersection_loss(points1, points2, line, rng, pool=False,
                                           mode=_scan_mode(mode), chunk=chunk)
    return loss, info[:, 0] > 0


def _scan_mode(mode):
    import os
    mode = mode or os.environ.get("RRL_SCAN_MODE", "cull")
    if mode not in ("strict", "lazy", "auto", "cull"):
        raise ValueError("mode must be 'cull', 'auto', 'strict' or 'lazy'")
    return mode


def chamfer_dist(points_x, points_y):
    """Symmetric Chamfer monitor: mean over all B*(M+N) nearest squared distances
    (code/loss.py:236-252).  Scalar tensor on points_x's device, differentiable."""
    return _ops.chamfer(points_x, points_y)


# ------------------------------------------------------------------------------- sampler
def generate_bbox(vertices):
    """(B, V, 3) -> (B, 8, 3) AABB corners, corner 0 = max, corner 7 = min
    (code/loss.py:325-351; same order as libigl's bounding_box).  Returned on the CPU like
    the reference's `torch.zeros(...)` buffer."""
    bb = _ops.aabb(vertices)
    mn, mx = bb[:, :3], bb[:, 3:]
    pick = torch.tensor([[1, 1, 1], [1, 1, 0], [1, 0, 1], [1, 0, 0], [0, 1, 1], [0, 1, 0],
                         [0, 0, 1], [0, 0, 0]], dtype=torch.bool, device=bb.device)
    return torch.where(pick[None], mx[:, None, :], mn[:, None, :]).cpu()


_pinned = {}


def _uniform_rounds(B, n, rounds, device_rng=None, dev=None):
    """The reference's CPU RNG stream: per round four `torch.rand(B, n)` draws in the order
    alpha1, u1, alpha2, u2 (code/loss.py:394-402), from torch's default CPU generator, so a
    `torch.manual_seed(s)` before the call selects the same candidates as in the reference.
    One `torch.rand` of the stacked shape consumes the generator exactly like the 4 * rounds
    separate calls (checked in tests/test_host.py) and lands in a reused pinned buffer, so the
    upload is one asynchronous copy; the result is on the GPU.  device_rng: a torch.device -> None is returned and the
    sampler kernels draw the uniforms themselves (the library's counter-based generator, rrl_hip.ops.sampler_rng: same
    distribution, a different stream -- for training loops and captured steps, where 4 * rounds * B * n host-generated
    floats per step would cost more than the loss itself)."""
    if device_rng is not None:
        return None  # drawn inside the sampler kernels by the library's own generator (rrl_hip.ops.sampler_rng)
    dev = dev if dev is not None else _ops.require_gpu()
    if B * n < 16:  # torch's scalar path for tiny tensors: keep the reference's call pattern
        return torch.stack([torch.stack([torch.rand(B, n) for _ in range(4)]) for _ in range(rounds)]).to(dev)
    key = (rounds, B, n, dev.index)
    slot = _pinned.get(key)
    if slot is None:
        if len(_pinned) > 8:
            _pinned.clear()
        slot = _pinned[key] = (torch.empty(rounds, 4, B, n).pin_memory(), torch.cuda.Event())
    else:
        slot[1].synchronize()  # the previous upload from this buffer has left the host
    buf, ev = slot
    torch.rand(rounds, 4, B, n, out=buf)
    with torch.cuda.device(dev):
        out = buf.to(dev, non_blocking=True)
        ev.record()
    return out


def Random_uniform_distribution_lines_batch_efficient(r, centers, N, device='cpu', *, device_rng=False):
    """One round of candidate lines: chords between two uniform points of the radius-r sphere
    around `centers` (code/loss.py:384-412).  (B, N, 6) = [unit direction, x0]."""
    B = r.shape[0]
    rands = _uniform_rounds(B, N, 1, _ops.require_gpu() if device_rng else None)
    lines, _ = _sample(rands, r, centers, None, None, shape=(1, B, N))
    return lines.to(device)


def Random_uniform_distribution_lines_batch_efficient_resample(r, centers, N, vertices1, vertices2,
                                                               device='cpu', *, rounds=10, device_rng=False,
                                                               out=None, box2=None, box1=None):
    """`rounds` (reference: 10) rejection rounds: a candidate is kept when it crosses the AABB
    of BOTH clouds by the reference's 12-triangle sub-area test; kept candidates fill an
    (B, N, 6) buffer front to back, overflow is dropped, unfilled rows stay all-zero
    (code/loss.py:415-432, 365-381).
    Keyword-only extras for loops that call this every step: out = a (B, N, 6) fp32 GPU tensor to
    fill in place, box2 = the (B, 6) AABB of `vertices2` from rrl_hip.ops.aabb when that cloud does
    not move (box1: the same for `vertices1`, e.g. from rrl_hip.ops.rigid_apply_aabb_into)."""
    B = r.shape[0]
    dev = _ops._home(out, vertices1, vertices2)  # the lines are built where the clouds live
    rands = _uniform_rounds(B, N, rounds, dev if device_rng else None, dev)
    bb2 = box2 if box2 is not None else _ops.aabb(vertices2)
    bb1 = box1 if box1 is not None else _ops.aabb(vertices1)
    lines, _ = _sample(rands, r, centers, bb1, bb2, out, shape=(rounds, B, N))
    return lines if out is not None else lines.to(device)


def _sample(rands, r, centers, bb1, bb2, out=None, shape=None):
    B = shape[1]
    rr = r.reshape(B, -1)[:, 0]
    return _ops.sample_lines(rands, rr, centers.reshape(B, 3), bb1, bb2, out, rng_shape=shape)


# ----------------------------------------------------------------- rigid transform module
class Reconstruction_point(nn.Module):
    """Single-pair rigid transform parameterised by one se(3) 6-vector `parameters_`
    (code/loss.py:437-463).  forward(points (N,3), points_neighbors (1,3N,3)) returns the
    transformed (N,3) cloud and (N,9) pseudo-triangles, `points @ R + T` (row-vector
    convention), computed by the HIP rigid-apply kernel."""

    def __init__(self, rotation=None, translation=None):
        super().__init__()
        if rotation is None or translation is None:
            axis = np.random.randn(3)  # numpy RNG, as code/loss.py:441-447
            axis = axis / np.linalg.norm(axis)
            trans = np.random.randn(3) * 0.001
            init = torch.from_numpy(np.concatenate([0.001 * axis, trans], 0).astype(np.float32))
        else:
            T = torch.zeros(4, 4)
            T[:3, :3] = rotation.reshape(3, 3)
            T[:3, 3] = translation.reshape(3)
            init = se3.log(T).reshape(-1) + torch.rand(6) * 0.6  # code/loss.py:449-453
        self.parameters_ = nn.Parameter(init)

    def Transform(self):
        if self.parameters_.is_cuda:  # one launch each way instead of ~40 / ~100 torch kernels
            return _ops.se3_exp(self.parameters_)
        return se3.exp3(self.parameters_)

    def forward(self, points, points_neighbors):
        R, T = self.Transform()
        moved = _ops.rigid_apply(points.reshape(1, -1, 3), R, T)
        moved_nb = _ops.rigid_apply(points_neighbors.reshape(1, -1, 3), R, T)
        return moved.reshape(-1, 3), moved_nb.reshape(-1, 9)


def Sample_neighs(points, num_sample=5000, num_neigh=3, device='cpu'):
    """Pseudo-triangle builder (code/loss.py:473-485): farthest-point sample <= num_sample
    points, then each with its num_neigh nearest neighbours -> (3*S, 3) rows [p, nn1, nn2].
    GPU farthest-point sampling + brute-force 3-NN (rrl_hip.neighbors); the FPS start index is
    drawn with torch.randint from the CPU generator like the reference's."""
    from rrl_hip import neighbors
    return neighbors.sample_neighs(points, num_sample, num_neigh)
