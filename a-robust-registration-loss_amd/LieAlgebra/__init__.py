"""se(3)/so(3) exp/log maps -- the subset of the reference's `LieAlgebra` package that the
loss path uses (code/loss.py:6 does `from LieAlgebra import *`, reaching se3.exp3 / se3.log
at code/loss.py:453,456).  Host-side torch on (B, 6) parameters: a few dozen flops per
sample, differentiable by autograd; the heavy work (rigid apply) is a HIP kernel."""
from . import sinc, so3, se3  # noqa: F401
