"""Direct RCCL binding for the per-step shard-payload all-reduce (one process per GPU, xGMI).

torch.distributed's all_reduce costs ~50 us of host time per call (work objects, event bookkeeping):
more than a third of the 0.1 ms step it follows, which makes a multi-GPU step host-bound.  The
payload is 14 floats and the pattern never changes, so this module talks to librccl itself: one
communicator (unique id from rank 0, shared through the existing process group), a private HIP
stream, `ncclAllReduce` on a double-used 14-float buffer.  Per step: a 56-byte copy kernel on the
compute stream, two event edges, one ncclAllReduce call -- a few microseconds of host time, and the
collective overlaps the next step's kernels like rrl_hip.dist.PayloadReducer (same interface).
Anything that fails during set-up falls back to that class.
"""
import ctypes
import os

import torch
import torch.distributed as dist

from .dist import PayloadReducer

_NCCL_FLOAT32, _NCCL_SUM = 7, 0


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]


def _load_rccl():
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")  # the instance torch loaded
    lib = ctypes.CDLL(path if os.path.exists(path) else "librccl.so")
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
    lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_void_p, ctypes.c_void_p]
    lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllReduce, lib.ncclCommDestroy):
        f.restype = ctypes.c_int
    return lib


class RcclReducer:
    """submit(payload): copy + asynchronous sum all-reduce of the 14 floats on a private stream,
    overlapping whatever the compute stream does next; finish(): the last submitted step's sums."""

    def __init__(self, device):
        self.lib = _load_rccl()
        rank, world = dist.get_rank(), dist.get_world_size()
        uid = _UniqueId()
        if rank == 0 and self.lib.ncclGetUniqueId(ctypes.byref(uid)) != 0:
            raise RuntimeError("ncclGetUniqueId failed")
        t = torch.frombuffer(bytearray(bytes(uid.internal)), dtype=torch.uint8).clone().to(device)
        dist.broadcast(t, src=0)  # the id travels over the existing process group
        ctypes.memmove(ctypes.byref(uid), bytes(t.cpu().numpy().tobytes()), 128)
        self.comm = ctypes.c_void_p()
        with torch.cuda.device(device):
            if self.lib.ncclCommInitRank(ctypes.byref(self.comm), world, uid, rank) != 0:
                raise RuntimeError("ncclCommInitRank failed")
        self.stream = torch.cuda.Stream(device=device)
        self.buf = torch.zeros(16, dtype=torch.float32, device=device)
        self.out = torch.zeros(14, dtype=torch.float32, device=device)
        self.ready = torch.cuda.Event()   # buf holds the new payload (compute stream)
        self.done = torch.cuda.Event()    # the all-reduce has finished with buf (private stream)
        self.pending = False
        # double-buffered variant for captured steps (see run_graphed)
        self.bufs = [torch.zeros(16, dtype=torch.float32, device=device) for _ in range(2)]
        self.readys = [torch.cuda.Event() for _ in range(2)]
        self.dones = [torch.cuda.Event() for _ in range(2)]
        self.pendings = [False, False]
        self.last = None

    def _allreduce(self, buf):
        rc = self.lib.ncclAllReduce(ctypes.c_void_p(buf.data_ptr()), ctypes.c_void_p(buf.data_ptr()), 14,
                                    _NCCL_FLOAT32, _NCCL_SUM, self.comm, ctypes.c_void_p(self.stream.cuda_stream))
        if rc != 0:
            raise RuntimeError(f"ncclAllReduce failed ({rc})")

    def allreduce_inline(self, t):
        """In-place sum all-reduce of the 14-float tensor `t` on the CURRENT stream -- called inside
        the captured step, so the collective becomes a node of the hipGraph: no events, no second
        stream (an event edge per step costs ~10 us of GPU time here: barrier packets with
        system-scope cache maintenance), only the collective's own device latency."""
        rc = self.lib.ncclAllReduce(ctypes.c_void_p(t.data_ptr()), ctypes.c_void_p(t.data_ptr()), 14, _NCCL_FLOAT32,
                                    _NCCL_SUM, self.comm,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f"ncclAllReduce failed ({rc})")
        return t

    def run_graphed(self, parity, replay):
        """One captured step whose LAST node copies its payload into self.bufs[parity]: no eager
        kernel sits between two graph launches (that alone cost ~15 us per step), only event
        edges -- the step two launches back must have left the buffer, the reduction waits for
        this launch."""
        cur = torch.cuda.current_stream()
        if self.pendings[parity]:
            cur.wait_event(self.dones[parity])
        replay()
        self.readys[parity].record(cur)
        self.stream.wait_event(self.readys[parity])
        self._allreduce(self.bufs[parity])
        self.dones[parity].record(self.stream)
        self.pendings[parity] = True
        self.last = parity

    def submit(self, payload):
        cur = torch.cuda.current_stream()
        if self.pending:
            cur.wait_event(self.done)     # the previous reduction is out of buf
        self.buf[:14].copy_(payload)
        self.ready.record(cur)
        self.stream.wait_event(self.ready)
        self._allreduce(self.buf)
        self.done.record(self.stream)
        self.pending = True
        self.last = None

    def finish(self):
        cur = torch.cuda.current_stream()
        for p in range(2):
            if self.pendings[p]:
                cur.wait_event(self.dones[p])
                self.pendings[p] = False
        if self.last is not None:
            self.out.copy_(self.bufs[self.last][:14])
            self.last = None
        elif self.pending:
            cur.wait_event(self.done)
            self.out.copy_(self.buf[:14])
        self.pending = False
        return self.out

    def close(self):
        if self.comm:
            torch.cuda.synchronize()
            self.lib.ncclCommDestroy(self.comm)
            self.comm = ctypes.c_void_p()


def make_reducer(device):
    """RcclReducer when a NCCL(=RCCL) process group is up and the direct binding initialises,
    else the torch.distributed based PayloadReducer."""
    if dist.is_initialized() and dist.get_backend() == "nccl" and os.environ.get("RRL_DIRECT_RCCL", "1") != "0":
        try:
            return RcclReducer(device)
        except Exception as exc:  # any set-up problem: keep the portable path
            import sys
            print(f"[rrl_hip.rccl] direct RCCL unavailable ({type(exc).__name__}: {exc}); using torch.distributed",
                  file=sys.stderr)
    return PayloadReducer(device)
