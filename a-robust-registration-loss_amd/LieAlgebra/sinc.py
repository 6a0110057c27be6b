"""sinc-type series used by the exponential map (reference: code/LieAlgebra/sinc.py:5-17,
91-103, 120-132).  For |t| < 0.01 the same Taylor polynomials as the reference are used,
elsewhere the closed forms.  Written with torch.where on a guarded argument so both value
and gradient are finite at t = 0."""
import torch

_EPS = 0.01


def _split(t):
    small = t.abs() < _EPS
    safe = torch.where(small, torch.ones_like(t), t)  # closed form never sees |t| < 0.01
    return small, safe, t * t


def sinc1(t):
    """sin(t) / t"""
    small, s, t2 = _split(t)
    series = 1 - t2 / 6 * (1 - t2 / 20 * (1 - t2 / 42))
    return torch.where(small, series, torch.sin(s) / s)


def sinc2(t):
    """(1 - cos t) / t^2"""
    small, s, t2 = _split(t)
    series = 1 / 2 * (1 - t2 / 12 * (1 - t2 / 30 * (1 - t2 / 56)))
    return torch.where(small, series, (1 - torch.cos(s)) / (s * s))


def sinc3(t):
    """(t - sin t) / t^3"""
    small, s, t2 = _split(t)
    series = 1 / 6 * (1 - t2 / 20 * (1 - t2 / 42 * (1 - t2 / 72)))
    return torch.where(small, series, (s - torch.sin(s)) / (s ** 3))
