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
    lib.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    lib.ncclCommUserRank.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    lib.ncclGetVersion.argtypes = [ctypes.POINTER(ctypes.c_int)]
    for f in (lib.ncclGetUniqueId, lib.ncclCommInitRank, lib.ncclAllReduce, lib.ncclCommDestroy,
              lib.ncclCommCount, lib.ncclCommUserRank, lib.ncclGetVersion):
        f.restype = ctypes.c_int
    return lib


def _all_ok(ok, device):
    """Collective AND over the ranks of the default process group: every rank takes the same branch
    afterwards (a set-up step that fails on one rank must fail on all, or the ranks end up issuing
    different collectives and hang)."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


class RcclReducer:
    """submit(payload): copy + asynchronous sum all-reduce of the 14 floats on a private stream,
    overlapping whatever the compute stream does next; finish(): the last submitted step's sums."""

    def __init__(self, device):
        """COLLECTIVE over the default process group: every rank must call it, and it either
        succeeds on all ranks or raises on all ranks (each local step that can fail is followed by
        an all-reduced success flag before the next collective is issued)."""
        rank, world = dist.get_rank(), dist.get_world_size()
        self.comm = ctypes.c_void_p()
        self.lib, err = None, None
        try:
            self.lib = _load_rccl()
        except (OSError, AttributeError) as exc:
            err = exc
        if not _all_ok(self.lib is not None, device):
            raise RuntimeError(f"librccl could not be bound on every rank ({err})")
        uid = _UniqueId()
        got = rank != 0 or self.lib.ncclGetUniqueId(ctypes.byref(uid)) == 0
        # the id travels over the existing process group together with rank 0's success flag: the
        # broadcast is issued by every rank whether or not ncclGetUniqueId worked
        t = torch.frombuffer(bytearray(bytes(uid.internal) + bytes([1 if got else 0])), dtype=torch.uint8).clone().to(device)
        dist.broadcast(t, src=0)
        raw = bytes(t.cpu().numpy().tobytes())
        if raw[128] != 1:
            raise RuntimeError("ncclGetUniqueId failed on rank 0")
        ctypes.memmove(ctypes.byref(uid), raw[:128], 128)
        with torch.cuda.device(device):
            rc = self.lib.ncclCommInitRank(ctypes.byref(self.comm), world, uid, rank)
        if not _all_ok(rc == 0, device):
            self.close()
            raise RuntimeError(f"ncclCommInitRank failed on some rank (here: {rc})")
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

    def evidence(self):
        """What the communicator itself reports: {"nranks", "rank", "version"} (ncclCommCount,
        ncclCommUserRank, ncclGetVersion) -- bench.py prints it so that a run record shows how many
        ranks RCCL saw."""
        n, r, v = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(-1)
        self.lib.ncclCommCount(self.comm, ctypes.byref(n))
        self.lib.ncclCommUserRank(self.comm, ctypes.byref(r))
        self.lib.ncclGetVersion(ctypes.byref(v))
        return {"nranks": n.value, "rank": r.value, "version": v.value}

    def close(self):
        if self.comm:
            torch.cuda.synchronize()
            self.lib.ncclCommDestroy(self.comm)
            self.comm = ctypes.c_void_p()


def make_reducer(device):
    """RcclReducer when a NCCL(=RCCL) process group is up and the direct binding initialises ON
    EVERY RANK, else the torch.distributed based PayloadReducer on every rank.  Collective: all
    ranks must call it.  RRL_DIRECT_RCCL=0 (set identically on all ranks) skips the binding."""
    if dist.is_initialized() and dist.get_backend() == "nccl" and os.environ.get("RRL_DIRECT_RCCL", "1") != "0":
        try:
            return RcclReducer(device)   # raises on all ranks or on none
        except RuntimeError as exc:
            import sys
            print(f"[rrl_hip.rccl] direct RCCL unavailable ({exc}); using torch.distributed", file=sys.stderr)
    return PayloadReducer(device)


def agree(ok, device):
    """all-ranks AND of a local success flag (True without a process group): lets a caller choose
    between two collective patterns -- e.g. the in-graph all-reduce vs the overlapped one -- the
    same way on every rank."""
    if not dist.is_initialized():
        return bool(ok)
    return _all_ok(ok, device)
