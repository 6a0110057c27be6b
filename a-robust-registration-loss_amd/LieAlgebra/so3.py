"""so(3): hat map, Rodrigues exponential, logarithm (reference: code/LieAlgebra/so3.py:17-27,
62-75, 95-131, 168-184)."""
import torch

from .sinc import sinc1, sinc2, sinc3  # noqa: F401


def mat(x):
    """[*, 3] -> [*, 3, 3] skew-symmetric hat matrix."""
    v = x.reshape(-1, 3)
    a, b, c = v[:, 0], v[:, 1], v[:, 2]
    z = torch.zeros_like(a)
    X = torch.stack([z, -c, b, c, z, -a, -b, a, z], dim=1)
    return X.reshape(*x.shape[:-1], 3, 3)


def vec(X):
    """inverse of mat: [*, 3, 3] -> [*, 3]"""
    M = X.reshape(-1, 3, 3)
    return torch.stack([M[:, 2, 1], M[:, 0, 2], M[:, 1, 0]], dim=1).reshape(*X.shape[:-2], 3)


def _angle(w):
    # |w| with a finite gradient at w = 0 (the reference's norm() yields NaN there)
    n2 = (w * w).sum(dim=1)
    safe = torch.where(n2 > 0, n2, torch.ones_like(n2))
    return torch.where(n2 > 0, safe.sqrt(), torch.zeros_like(n2)).reshape(-1, 1, 1)


def exp(x):
    """Rodrigues: R = I + sinc1(t) W + sinc2(t) W^2"""
    w = x.reshape(-1, 3)
    t = _angle(w)
    W = mat(w)
    S = W.bmm(W)
    I = torch.eye(3, dtype=w.dtype, device=w.device)
    R = I + sinc1(t) * W + sinc2(t) * S
    return R.reshape(*x.shape[:-1], 3, 3)


def inverse(g):
    return g.transpose(-1, -2)


def log(g):
    """SO(3) -> so(3) vector.  Near theta = pi the axis comes from the diagonal of
    (R + I) * theta^2 / 2 with signs fixed by the off-diagonal terms."""
    R = g.reshape(-1, 3, 3)
    tr = R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]
    t = torch.acos((tr - 1) / 2)
    sc = sinc1(t)
    regular = sc.abs() > 1.0e-7
    X = torch.zeros_like(R)
    if regular.any():
        Rr = R[regular]
        X[regular] = (Rr - Rr.transpose(1, 2)) / (2 * sc[regular].reshape(-1, 1, 1))
    if (~regular).any():
        Rs = R[~regular]
        A = (Rs + torch.eye(3, dtype=R.dtype, device=R.device)) * (t[~regular] ** 2).reshape(-1, 1, 1) / 2
        s3 = torch.sign(A[:, 0, 2])
        s3[s3 == 0] = 1
        s23 = torch.sign(A[:, 1, 2])
        s23[s23 == 0] = 1
        w = torch.stack([A[:, 0, 0].sqrt(), A[:, 1, 1].sqrt() * (s23 * s3), A[:, 2, 2].sqrt() * s3], dim=-1)
        X[~regular] = mat(w)
    return vec(X.reshape(g.shape))


def inv_vecs_Xg_ig(x):
    """H = V^-1 with V = I + sinc2 W + sinc3 W^2: I - W/2 + eta W^2."""
    w = x.reshape(-1, 3)
    t = _angle(w)
    W = mat(w)
    S = W.bmm(W)
    I = torch.eye(3, dtype=w.dtype, device=w.device)
    small = t < 0.01
    safe = torch.where(small, torch.ones_like(t), t)
    t2 = t * t
    series = ((t2 / 40 + 1) * t2 / 42 + 1) * t2 / 720 + 1 / 12
    closed = (1 - (safe / 2) / torch.tan(safe / 2)) / (safe * safe)
    eta = torch.where(small, series, closed)
    H = I - 0.5 * W + eta * S
    return H.reshape(*x.shape[:-1], 3, 3)
