"""rrl_hip -- host side of the MI355X intersected-line registration loss.

  build      compiles librrl_hip.so (hipcc, gfx950)
  _lib       ctypes binding of the C ABI (include/rrl.h)
  ops        torch.autograd front-ends (loss, rigid apply, chamfer, sampler)
  dist       batch-shard over ranks + RCCL all-reduce
  synth      seeded synthetic pairs for tests and bench
The reference-compatible import surface is ../loss.py, ../utils.py, ../LieAlgebra.
"""
from ._lib import RRLError  # noqa: F401
