"""se(3) exponential / logarithm (reference: code/LieAlgebra/se3.py:57-106, 124-147).
A twist is xi = (w, v): rotation vector first, then the translational part."""
import torch

from . import so3
from .sinc import sinc1, sinc2, sinc3


def _exp_parts(x):
    xi = x.reshape(-1, 6)
    w, v = xi[:, 0:3], xi[:, 3:6]
    t = so3._angle(w)
    W = so3.mat(w)
    S = W.bmm(W)
    I = torch.eye(3, dtype=xi.dtype, device=xi.device)
    R = I + sinc1(t) * W + sinc2(t) * S
    V = I + sinc2(t) * W + sinc3(t) * S
    p = V.bmm(v.reshape(-1, 3, 1))
    return R, p


def exp3(x):
    """[*, 6] -> (R [n,3,3], p [n,3]) -- what Reconstruction_point.Transform returns."""
    R, p = _exp_parts(x)
    return R, p.reshape(-1, 3)


def exp(x):
    """[*, 6] -> [*, 4, 4] homogeneous matrix."""
    R, p = _exp_parts(x)
    n = R.shape[0]
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=R.dtype, device=R.device).expand(n, 1, 4)
    g = torch.cat([torch.cat([R, p], dim=2), bottom], dim=1)
    return g.reshape(*x.shape[:-1], 4, 4)


def log(g):
    """[*, 4, 4] -> [*, 6]"""
    G = g.reshape(-1, 4, 4)
    R, p = G[:, 0:3, 0:3], G[:, 0:3, 3]
    w = so3.log(R)
    v = so3.inv_vecs_Xg_ig(w).bmm(p.reshape(-1, 3, 1)).reshape(-1, 3)
    return torch.cat([w, v], dim=1).reshape(*g.shape[:-2], 6)


def inverse(g):
    G = g.reshape(-1, 4, 4)
    Q = G[:, 0:3, 0:3].transpose(1, 2)
    q = -Q.bmm(G[:, 0:3, 3:4])
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=G.dtype, device=G.device).expand(G.shape[0], 1, 4)
    return torch.cat([torch.cat([Q, q], dim=2), bottom], dim=1).reshape(g.shape)


def transform(g, a):
    """Applies g (.., 4, 4) to points a: (.., 3, N) when ranks match, else (.., N, 3) per point
    (reference: code/LieAlgebra/se3.py:137-147)."""
    G = g.reshape(-1, 4, 4)
    R = G[:, 0:3, 0:3].reshape(*g.shape[:-2], 3, 3)
    p = G[:, 0:3, 3].reshape(*g.shape[:-2], 3)
    if g.dim() == a.dim():
        return R.matmul(a) + p.unsqueeze(-1)
    return R.matmul(a.unsqueeze(-1)).squeeze(-1) + p
