"""Pseudo-triangle builder on the GPU (reference: Sample_neighs, code/loss.py:473-485 with
utils.farthest_point_sample, code/utils.py:275-296, and sklearn's KDTree).  SURVEY.md §8f row 1."""
import numpy as np
import torch

from .ops import _home, _p, _run


def fps(points, num_sample, start=None):
    """points (B, n, 3) tensor -> (B, S) int32 indices in farthest-point order.  `start` (B,)
    defaults to torch.randint(0, n, (B,)) from the CPU generator, like the reference."""
    dev = _home(points)
    pts = points.detach().to(device=dev, dtype=torch.float32).contiguous()
    B, n, _ = pts.shape
    S = min(int(num_sample), n)
    if start is None:
        start = torch.randint(0, n, (B,), dtype=torch.long)
    st = start.to(device=dev, dtype=torch.int32).contiguous()
    out = torch.empty(B, S, dtype=torch.int32, device=dev)
    scratch = torch.empty(B, n, dtype=torch.float32, device=dev)
    _run(dev, "rrl_fps", _p(pts), _p(st), _p(out), _p(scratch), B, n, S)
    return out


def knn3(points, query_idx):
    """points (B, n, 3), query_idx (B, S) -> (B, S, 3) int32: the 3 nearest points (itself first)."""
    dev = _home(points, query_idx)
    pts = points.detach().to(device=dev, dtype=torch.float32).contiguous()
    qi = query_idx.to(device=dev, dtype=torch.int32).contiguous()
    B, n, _ = pts.shape
    S = qi.shape[1]
    nn = torch.empty(B, S, 3, dtype=torch.int32, device=dev)
    _run(dev, "rrl_knn3", _p(pts), _p(qi), _p(nn), B, n, S)
    return nn


def sample_neighs(points, num_sample=5000, num_neigh=3):
    """numpy (n, 3) -> numpy (3 S, 3): rows [p, nn1, nn2] of the S farthest-point samples
    (S = min(num_sample, n)), the row layout every caller reshapes to (S, 9)."""
    if num_neigh != 3:
        raise ValueError("the loss uses pseudo-triangles: num_neigh must be 3")
    pts_np = np.asarray(points)
    pts = torch.from_numpy(np.ascontiguousarray(pts_np, dtype=np.float32))[None]
    idx = fps(pts, num_sample)
    nn = knn3(pts, idx)[0].long().cpu().numpy()
    out = pts_np[nn.reshape(-1)]  # gathers from the caller's array: keeps its dtype
    return out.reshape(-1, 3)
