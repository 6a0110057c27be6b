"""Seeded synthetic registration pairs.

Per sample: N points on a bumpy ellipsoid; the target is an independent M-sample
of the same surface, rotated 20 degrees about a random axis, translated by
(0.05, -0.03, 0.02) and perturbed by N(0, 0.01^2) noise; both clouds centred;
pseudo-triangles are each point with its two nearest neighbours (row layout
[P0 P1 P2], the output contract of the reference's Sample_neighs,
code/loss.py:473-485).  Host-side numpy only: this is workload generation, not
part of the measured path.
"""
import numpy as np


def _surface(rng, n):
    u = rng.standard_normal((n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    rad = 1.0 + 0.25 * np.sin(3.0 * u[:, 0]) * np.cos(2.0 * u[:, 1])
    return (u * rad[:, None]) * np.array([1.0, 0.7, 0.5])


def _rotation(axis, deg):
    axis = axis / np.linalg.norm(axis)
    a = np.deg2rad(deg)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)


def knn_triangles(points, k=3):
    """(n,3) float32 -> (n, 3*k) rows [P, nn1, nn2] (self is neighbour 0), the row layout of the
    reference's Sample_neighs.  scipy cKDTree on the host: workload generation only."""
    from scipy.spatial import cKDTree
    p = np.ascontiguousarray(points, np.float32)
    _, idx = cKDTree(p).query(p, k=k)
    idx[:, 0] = np.arange(p.shape[0])  # duplicates: keep the point itself first
    return p[idx].reshape(p.shape[0], 3 * k)


def make_pair(seed, n, m, crop=False, noise=0.01, rot_deg=20.0):
    """Returns dict(src (n,3), tar (m,3), src_tri (n,9), tar_tri (m,9), center (3,), radius).
    radius = 0.5 * AABB diagonal of the target (DCP/FMR convention,
    dcp/Train_DCP.py:234-236); center = target mean."""
    rng = np.random.default_rng(seed)
    src = _surface(rng, n)
    m_gen = 2 * m if crop else m
    tar = _surface(rng, m_gen)
    axis = rng.standard_normal(3)
    tar = tar @ _rotation(axis, rot_deg).T + np.array([0.05, -0.03, 0.02])
    tar = tar + noise * rng.standard_normal(tar.shape)
    if crop:  # keep the half-space holding 50 % of the points (C4: partial overlap)
        nrm = rng.standard_normal(3)
        proj = tar @ (nrm / np.linalg.norm(nrm))
        tar = tar[np.argsort(proj, kind="stable")[:m]]
    src = (src - src.mean(0)).astype(np.float32)
    tar = (tar - tar.mean(0)).astype(np.float32)
    diag = float(np.linalg.norm(tar.max(0) - tar.min(0)))
    return dict(src=src, tar=tar, src_tri=knn_triangles(src), tar_tri=knn_triangles(tar),
                center=tar.mean(0).astype(np.float32), radius=np.float32(0.5 * diag))


def uniform_streams(seed, rounds, n_lines):
    """(rounds, 4, n_lines) uniform [0,1) draws: the four CPU torch.rand streams of
    code/loss.py:394-402 (alpha1, u1, alpha2, u2), from torch's CPU generator so a
    given seed reproduces what the reference sampler would consume."""
    import torch
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return torch.rand(rounds, 4, n_lines, generator=g).numpy()
