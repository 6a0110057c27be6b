"""torch.autograd front-ends of the HIP kernels (device memory + streams are PyTorch's;
all compute is librrl_hip.so).  No fallback path exists: CPU tensors are moved to the
current GPU, and a missing library / missing GPU raises RRLError.
"""
import ctypes
import os
import sys

import torch

from . import _lib
from ._lib import RRLError, check

SCAN_STRICT, SCAN_LAZY, SCAN_AUTO, SCAN_CULL = 0, 1, 2, 3
_MODES = {"strict": SCAN_STRICT, "lazy": SCAN_LAZY, "auto": SCAN_AUTO, "cull": SCAN_CULL}

# workspace fields, in the order of include/rrl.h's RRL_WS_* enum: (name, dtype, shape)
_WS_FIELDS = [
    ("status", torch.int32, lambda B, N, M, L, G: (4,)),
    ("nvals", torch.int32, lambda B, N, M, L, G: (B,)),
    ("nsel", torch.int32, lambda B, N, M, L, G: (B,)),
    ("pmax", torch.float32, lambda B, N, M, L, G: (2, B)),
    ("count1", torch.int32, lambda B, N, M, L, G: (B, L)),
    ("count2", torch.int32, lambda B, N, M, L, G: (B, L)),
    ("hit1", torch.int32, lambda B, N, M, L, G: (B, L, 4)),
    ("hit2", torch.int32, lambda B, N, M, L, G: (B, L, 4)),
    ("ptri1", torch.float32, lambda B, N, M, L, G: (B, N, 12)),
    ("ptri2", torch.float32, lambda B, N, M, L, G: (B, M, 12)),
    ("p0s1", torch.float32, lambda B, N, M, L, G: (B, (N + 63) // 64 * 64, 4)),
    ("p0s2", torch.float32, lambda B, N, M, L, G: (B, (M + 63) // 64 * 64, 4)),
    ("idx1", torch.int32, lambda B, N, M, L, G: (B, (N + 63) // 64 * 64)),
    ("idx2", torch.int32, lambda B, N, M, L, G: (B, (M + 63) // 64 * 64)),
    ("grp1", torch.float32, lambda B, N, M, L, G: (B, (N + 63) // 64, 13, 4)),  # sphere tree
    ("grp2", torch.float32, lambda B, N, M, L, G: (B, (M + 63) // 64, 13, 4)),
    ("crec1", torch.float32, lambda B, N, M, L, G: (B, (N + 15) // 16 * 16, 4)),
    ("crec2", torch.float32, lambda B, N, M, L, G: (B, (M + 15) // 16 * 16, 4)),
    ("apart", torch.float32, lambda B, N, M, L, G: (2, B, (max(N, M) + 255) // 256, 8)),
    ("kj", torch.uint8, lambda B, N, M, L, G: (B, L)),
    ("sel", torch.int32, lambda B, N, M, L, G: (B, L)),
    ("hs1", torch.int32, lambda B, N, M, L, G: (B, L, 4)),
    ("hs2", torch.int32, lambda B, N, M, L, G: (B, L, 4)),
    ("w1", torch.float32, lambda B, N, M, L, G: (B, L, 4, 3)),
    ("w2", torch.float32, lambda B, N, M, L, G: (B, L, 4, 3)),
    ("Q1", torch.float32, lambda B, N, M, L, G: (B, L, 4, 4)),
    ("Q2", torch.float32, lambda B, N, M, L, G: (B, L, 4, 4)),
    ("D", torch.float32, lambda B, N, M, L, G: (B, L, 16)),
    ("vals", torch.float32, lambda B, N, M, L, G: (B, (L + 1023) // 1024 * 1024, 16)),
    ("med", torch.float32, lambda B, N, M, L, G: (G,)),
    ("bcnt", torch.int32, lambda B, N, M, L, G: (G, 16)),
    ("bsum", torch.int64, lambda B, N, M, L, G: (G, 16, 2)),
    ("info", torch.int32, lambda B, N, M, L, G: (G, 4)),
    ("tri1t", torch.float32, lambda B, N, M, L, G: (B, N, 9)),
    ("g1", torch.float32, lambda B, N, M, L, G: (B, N, 9)),
    ("rpart", torch.float32, lambda B, N, M, L, G: (B, (3 * N + 1023) // 1024 + 1, 12)),
    ("gacc", torch.float32, lambda B, N, M, L, G: (12 * B + 16,)),
    ("kjc", torch.uint8, lambda B, N, M, L, G: (B, (L + 1023) // 1024 * 1024)),
    ("blkcnt", torch.int32, lambda B, N, M, L, G: (B * ((L + 1023) // 1024 + 1),)),
    ("histg", torch.int32, lambda B, N, M, L, G: (2 * B * 2 * 4096 if max(N, M) > 4096 else 4,)),
    ("del1", torch.float32, lambda B, N, M, L, G: (B, N)),
    ("del2", torch.float32, lambda B, N, M, L, G: (B, M)),
    ("mhist", torch.int32, lambda B, N, M, L, G: (B, 2048)),
    ("mctl", torch.int32, lambda B, N, M, L, G: (B, 64)),
    ("msum", torch.int64, lambda B, N, M, L, G: (B, 32)),
    ("mcand", torch.int32, lambda B, N, M, L, G: (B, 2048)),
    ("lmax", torch.float32, lambda B, N, M, L, G: (B, 64, 2)),
    ("lidc", torch.int32, lambda B, N, M, L, G: (B, (L + 1023) // 1024 * 1024)),
    ("vlist", torch.float32, lambda B, N, M, L, G: (B, (L + 1023) // 1024, 16384)),
    ("vlcnt", torch.int32, lambda B, N, M, L, G: (B * ((L + 1023) // 1024 + 1),)),
    ("chain", torch.int32, lambda B, N, M, L, G: (B, 4)),
    ("gfix", torch.int64, lambda B, N, M, L, G: (B * (N + M) * 9 + B,)),
]
_layout_cache = {}
_FIELD_INDEX = {name: i for i, (name, _, _) in enumerate(_WS_FIELDS)}
_ITEMSIZE = {torch.int32: 4, torch.float32: 4, torch.uint8: 1, torch.int64: 8}
_spec_cache = {}  # (dims) -> {field: (offset, nbytes, dtype, shape)}


_gpu_ok = False


def require_gpu():
    global _gpu_ok
    if not _gpu_ok:
        if not torch.cuda.is_available():
            raise RRLError("no MI355X visible (torch.cuda.is_available() is False); "
                           "this package has no CPU fallback")
        _gpu_ok = True
    return torch.device("cuda", torch.cuda.current_device())


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


# ---- torch internals this module uses for SPEED, each with a public-API fallback (round 5) ---------
#   torch._C._cuda_getCurrentRawStream        raw hipStream_t without the ~15 us Stream wrapper -> torch.cuda.current_stream
#   torch._C._autograd._unsafe_set_version_counter   bump a tensor's version after a raw-pointer write -> the module's OWN
#                                             write map (_raw_writes), which every cache key of this module includes anyway
#   Tensor._base                              recognising the trainers' [j:j+1] loop -> without it the per-call path runs
# INTERNALS says which fast paths are live; a missing symbol degrades with ONE warning, never an AttributeError.
_raw_stream_fn = getattr(getattr(torch, "_C", None), "_cuda_getCurrentRawStream", None)
_set_version_fn = getattr(getattr(getattr(torch, "_C", None), "_autograd", None), "_unsafe_set_version_counter", None)
INTERNALS = {"raw_stream": _raw_stream_fn is not None, "version_bump": _set_version_fn is not None,
             "tensor_base": hasattr(torch.Tensor, "_base"), "torch": torch.__version__}
_warned = set()


def _warn_once(what):
    if what not in _warned:
        _warned.add(what)
        import warnings
        warnings.warn(f"rrl_hip: torch {torch.__version__} lacks {what}; using the public-API fallback (slower, same results)")


def _raw_stream(index):
    """hipStream_t (int) of torch's current stream on device `index`."""
    if _raw_stream_fn is not None:
        return _raw_stream_fn(index)
    _warn_once("torch._C._cuda_getCurrentRawStream")
    return torch.cuda.current_stream(index).cuda_stream


def _stream(dev=None):
    # raw hipStream_t of torch's current stream ON THE DEVICE THE DATA LIVES ON (the Stream object
    # wrapper costs ~15 us per call)
    return ctypes.c_void_p(_raw_stream(dev.index if dev is not None else torch.cuda.current_device()))


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NOGUARD = _NoGuard()


def _guard(dev):
    """Context that makes `dev` the current HIP device for the C calls inside: a kernel must be
    launched on a stream of the device that owns its pointers.  Free when it already is."""
    if dev.index == torch.cuda.current_device():
        return _NOGUARD
    return torch.cuda.device(dev)


def _run(dev, name, *args):
    """One C entry point on `dev`'s current stream (every entry takes the stream last)."""
    with _guard(dev):
        check(getattr(_lib.load(), name)(*args, _stream(dev)), name)


_raw_writes = {}     # data_ptr -> serial of the library's latest raw-pointer write into that buffer
_raw_serial = [0]    # monotone: a pruned map never repeats a key
_raw_floor = [0]     # serial at the last prune: what a buffer missing from the map is keyed on (never the old default 0)


def _write_key(t):
    """What this module's caches key a tensor's CONTENT on: (data_ptr, torch's version counter, the library's own
    raw-write serial of that buffer).  The third part makes the caches independent of the private version-bump hook."""
    p = t.data_ptr()
    return (p, t._version, _raw_writes.get(p, _raw_floor[0]))


def _touched(*tensors):
    """The library wrote these caller-owned tensors through raw pointers: record the write in the module's own map (every
    cache key of this module -- the steps' kept target, the drop-in batch cache -- includes it, _write_key) and, where
    torch offers the hook, advance their autograd version counters like an in-place torch op would (autograd's own
    saved-tensor checks then see the change too)."""
    ts = [t for t in tensors if isinstance(t, torch.Tensor)]
    if not ts:
        return
    if len(_raw_writes) > 4096:  # bounded; the serial keeps growing and the floor moves with it, so old keys never come back
        _raw_writes.clear()
        _raw_floor[0] = _raw_serial[0]
    for t in ts:
        _raw_serial[0] += 1
        _raw_writes[t.data_ptr()] = _raw_serial[0]
    if _set_version_fn is not None:
        try:
            _set_version_fn(ts, [t._version + 1 for t in ts])
        except (TypeError, RuntimeError):
            pass
    else:
        _warn_once("torch._C._autograd._unsafe_set_version_counter")


def _home(*tensors):
    """The device an op runs on: that of its first GPU argument, else the current GPU."""
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    return require_gpu()


def _prep(t, name, last=None, dev=None):
    """fp32, contiguous, on the op's GPU `dev` (callers pass slices / reshaped views and CPU
    tensors; a tensor on another GPU is copied over)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if last is not None and t.shape[-1] != last:
        raise ValueError(f"{name}: last dimension must be {last}, got {tuple(t.shape)}")
    if t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and (dev is None or t.device == dev):
        return t.detach()  # the common case: nothing to convert
    if dev is None:
        dev = t.device if t.is_cuda else require_gpu()
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


def _layout(B, N, M, L):
    key = (B, N, M, L)
    if key not in _layout_cache:
        lib = _lib.load()
        offs = (ctypes.c_size_t * len(_WS_FIELDS))()
        check(lib.rrl_workspace_layout(B, N, M, L, offs), "rrl_workspace_layout")
        _layout_cache[key] = (int(lib.rrl_workspace_bytes(B, N, M, L)), [int(o) for o in offs])
    return _layout_cache[key]


class LossState:
    """One loss evaluation: the single device workspace (include/rrl.h RRL_WS_*) that carries
    every intermediate from forward to backward, plus loss[G].  Named fields are lazy views."""

    def __init__(self, B, N, M, L, G, dev):
        self.dims = (B, N, M, L, G)
        self.nbytes, self.offsets = _layout(B, N, M, L)
        self.ws = torch.empty(self.nbytes, dtype=torch.uint8, device=dev)
        self.loss = torch.empty(G, dtype=torch.float32, device=dev)

    def __getattr__(self, name):
        i = _FIELD_INDEX.get(name)
        if i is None:
            raise AttributeError(name)
        if name in ("count2", "hit2"):  # a carried-over target: its hit counts / lists live where its scan left them
            tgt = self.__dict__.get("target_state")
            if tgt is not None and tgt is not self:
                return getattr(tgt, name)
        specs = _spec_cache.get(self.dims)
        if specs is None:
            specs = _spec_cache[self.dims] = {}
        spec = specs.get(name)
        if spec is None:
            _, dtype, shape = _WS_FIELDS[i]
            shp = shape(*self.dims)
            n = 1
            for d in shp:
                n *= d
            spec = specs[name] = (self.offsets[i], n * _ITEMSIZE[dtype], dtype, shp)
        off, nb, dtype, shp = spec
        return self.ws[off:off + nb].view(dtype).reshape(shp)

    @property
    def nbuckets(self):
        return self.info[:, 0]


def _check_range(rng):
    s_m, s_n, e_m, e_n = (int(v) for v in rng)
    if not (1 <= s_m and 1 <= s_n and e_m <= 5 and e_n <= 5):
        raise ValueError("bucket range must lie within 1..4 (RRL_MAX_HITS), as every reference "
                         "caller's (1, 1, 5, 5) does")
    return s_m, s_n, e_m, e_n


def _target_ws(target_from, B, N, M, L):
    """Workspace pointer of an earlier evaluation whose target scan is carried over."""
    if target_from is None:
        return None
    if tuple(target_from.dims[:4]) != (B, N, M, L):
        raise ValueError(f"target_from was evaluated at {tuple(target_from.dims[:4])}, not {(B, N, M, L)}")
    # the workspace that HOLDS the target's scan (records, hit counts and lists): of the full evaluation at the end of a chain
    # of carried-over ones
    tgt = getattr(target_from, "target_state", None) or target_from
    if getattr(tgt, "counts_cleared", False):
        raise ValueError("target_from: that state's hit counts were cleared by a chained step (RRL_F_CHAIN); build the step "
                         "whose target is carried over with chain=False")
    return _p(tgt.ws)


_order_ws_bytes = {}  # (B, n) -> scratch bytes of rrl_cloud_order


def cloud_order(tri):
    """Spatial order of the clouds tri (B, n, 9) -- pseudo-triangles; only the first point of a row places it -- or of
    point clouds (B, n, 3) (the Chamfer monitor's inputs: ops.chamfer(x, y, order_x=, order_y=)), computed
    ONCE per cloud (include/rrl.h rrl_cloud_order): int32 (B, 64 ceil(n / 64)), sorted position -> triangle index.
    A rigid motion preserves the order, so the same tensor serves every pose of the cloud: hand it to
    RegistrationStep / intersection_loss / registration_loss (order1= / order2=) and the per-step cell sort
    disappears.  Any permutation gives identical labels and loss; this one makes the culled scan fast."""
    dev = _home(tri)
    if not isinstance(tri, torch.Tensor) or tri.dim() != 3 or tri.shape[-1] not in (3, 9):
        raise ValueError("cloud_order takes pseudo-triangles (B, n, 9) or points (B, n, 3)")
    t = _prep(tri, "tri", None, dev)
    B, n, _ = t.shape
    if n > 65536:
        raise ValueError("clouds beyond 65536 triangles are not sorted (the dense scan serves them)")
    order = torch.zeros(B, (n + 63) // 64 * 64, dtype=torch.int32, device=dev)
    if B == 0 or n == 0:
        return order
    # scratch per CALL from torch's caching allocator, which is stream-aware: two calls of one shape on two streams /
    # from two threads never sort in the same memory (a per-shape global buffer did); the call runs once per
    # cloud, so the allocation is not on any step's path
    nb = _order_ws_bytes.get((B, n))
    if nb is None:
        nb = _order_ws_bytes[(B, n)] = int(_lib.load().rrl_cloud_order_workspace_bytes(B, n))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    _run(dev, "rrl_cloud_order" if t.shape[-1] == 9 else "rrl_cloud_order_points", _p(t), _p(order), _p(ws), ws.numel(), B, n)
    return order


def _check_order(order, B, n, dev, name):
    if order is None:
        return None
    if not (isinstance(order, torch.Tensor) and order.dtype == torch.int32 and order.is_cuda and order.device == dev
            and order.is_contiguous() and tuple(order.shape) == (B, (n + 63) // 64 * 64)):
        raise ValueError(f"{name} must be the contiguous int32 (B, 64 ceil(n / 64)) = {(B, (n + 63) // 64 * 64)} tensor of "
                         f"ops.cloud_order on the op's GPU ({dev}); got "
                         + (f"{order.dtype} {tuple(order.shape)} on {order.device}" if isinstance(order, torch.Tensor) else repr(type(order))))
    if ORDER_DEBUG and n > 0:  # RRL_ORDER_DEBUG=1: a host-side permutation check (synchronises; debugging only)
        srt = torch.sort(order[:, :n].to(torch.int64), dim=1).values
        if not bool((srt == torch.arange(n, device=dev)).all()):
            raise ValueError(f"{name}: every row's first n entries must be a permutation of [0, n)")
    return order


ORDER_DEBUG = os.environ.get("RRL_ORDER_DEBUG", "0") == "1"


_REDUCE = {"auto": 0, "single": 1, "tiled": 2, "xchg": 3}


class ChamferRide:
    """include/rrl.h rrl_chamfer_rider: the Chamfer monitor of a loss evaluation (what chamfer_from_state computes after
    it), issued INSIDE the evaluation's culled-scan launch -- the walk needs the records launch only, and launches of one
    stream never overlap on this stack, so riding along hides its time beside the scan's.  Hand it to make_opts(chamfer=)
    (registration_loss / intersection_loss / RegistrationStep take chamfer=True and do that themselves); after the call
    `.done` says whether it rode (else: chamfer_from_state, which looks here first).  Buffers are this object's own."""

    def __init__(self, B, N, M, dev):
        nb = _chamfer_ws_bytes.get((B, N, M))
        if nb is None:
            nb = _chamfer_ws_bytes[(B, N, M)] = int(_lib.load().rrl_chamfer_workspace_bytes(B, N, M))
        self.dims = (B, N, M)
        self.ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        self.bx = torch.empty(B, N, dtype=torch.int64, device=dev)
        self.by = torch.empty(B, M, dtype=torch.int64, device=dev)
        self.val = torch.empty(1, device=dev)
        self.c = _lib.ChamferRider(_p(self.ws), nb, _p(self.bx), _p(self.by), _p(self.val), 0)

    @property
    def done(self):
        return bool(self.c.done)

    def arm(self):
        self.c.done = 0
        return self


def make_opts(order1=None, order2=None, target_kept=False, reduce_mode=None, deterministic=None, sort_parts=None,
              scan_variant=None, counters=None, chamfer=None, payload=None, problems=None, chain=0, chain_left=None):
    """include/rrl.h rrl_opts for one call (None = the library default everywhere); the returned object keeps the
    tensors it points at alive (.keep).  chain: _lib.F_CHAIN [| _lib.F_CHAINED] (chained steps, include/rrl.h), chain_left:
    the ctypes.c_int32 that receives "this call leaves the workspace chain-clean"."""
    if order1 is None and order2 is None and not target_kept and reduce_mode is None and deterministic is None \
            and sort_parts is None and scan_variant is None and counters is None and chamfer is None and payload is None \
            and not problems and not chain:
        return None
    o = _lib.Opts(flags=(_lib.F_TARGET_KEPT if target_kept else 0) | int(chain),
                  chain_left=ctypes.addressof(chain_left) if chain_left is not None else None,
                  reduce_mode=-1 if reduce_mode is None else _REDUCE.get(reduce_mode, reduce_mode),
                  deterministic=-1 if deterministic is None else int(bool(deterministic)),
                  sort_parts=-1 if sort_parts is None else int(sort_parts),
                  scan_variant=-1 if scan_variant is None else int(scan_variant),
                  order1=order1.data_ptr() if order1 is not None else None,
                  order2=order2.data_ptr() if order2 is not None else None,
                  scan_counters=counters.data_ptr() if counters is not None else None,
                  scan_counter_rows=counters.shape[0] if counters is not None else 0,
                  chamfer=ctypes.addressof(chamfer.c) if chamfer is not None else None,
                  payload=payload.data_ptr() if payload is not None else None, problems=int(problems or 0))
    o.keep = (order1, order2, counters, chamfer, payload, chain_left)  # (o.problems: the ctypes field itself)
    o.ride = chamfer  # a ChamferRide (or None): the forwards arm it before the call and leave it on the LossState when it rode
    return o


def _optr(opts):
    return ctypes.byref(opts) if opts is not None else None


def _arm_ride(opts):
    ride = getattr(opts, "ride", None)
    if ride is not None:
        ride.arm()
    return ride


def _keep_ride(st, ride):
    """After a forward into `st`: the Chamfer walk that rode in its scan launch (chamfer_from_state looks here first)."""
    st.cham_ride = ride if (ride is not None and ride.done) else None


def loss_forward_raw(tri1, tri2, line, rng=(1, 1, 5, 5), pool=False, mode="cull", chunk=0,
                     staged=False, target_from=None, opts=None, state=None):
    """Forward on already-prepared GPU tensors; returns the LossState.  staged=True issues the
    four stages through their individual C entry points instead of the fused one.
    target_from: LossState of an earlier call with the SAME tri2 and line -- its target scan is
    reused (rrl_loss_forward_cached).
    opts: make_opts(...) -- per-call options (prepared orders, reduce mode, ...); state: evaluate into this LossState
    (same shape) instead of a new one (needed for target_kept: the target's records live in the workspace)."""
    lib = _lib.load()
    B, N, _ = tri1.shape
    M, L = tri2.shape[1], line.shape[1]
    G = 1 if pool else B
    s_m, s_n, e_m, e_n = _check_range(rng)
    dev = tri1.device
    if tri2.device != dev or line.device != dev:
        raise ValueError("tri1, tri2 and line must live on the same GPU")
    if state is not None:
        if tuple(state.dims) != (B, N, M, L, G) or state.ws.device != dev:
            raise ValueError("state was made for another shape / device")
        st = state
    else:
        st = LossState(B, N, M, L, G, dev)
    st.target_state = getattr(target_from, "target_state", None) or target_from  # whose workspace holds cloud 2
    ws, nb = _p(st.ws), st.nbytes
    op = _optr(opts)
    ride = _arm_ride(opts)
    with _guard(dev):
        s = _stream(dev)
        if not staged:
            check(lib.rrl_loss_forward_ex(_p(tri1), _p(tri2), _p(line), ws, nb, _p(st.loss), B, N, M,
                                          L, s_m, s_n, e_m, e_n, int(pool), _MODES[mode], int(chunk),
                                          _target_ws(target_from, B, N, M, L), op, s),
                  "rrl_loss_forward")
            _keep_ride(st, ride)
            return st
        check(lib.rrl_tri_prepare_ex(_p(tri1), _p(tri2), ws, nb, B, N, M, L, op, s), "rrl_tri_prepare")
        if staged == "mixed":  # (tests) the later stages WITHOUT the build's options: the workspace describes itself
            op = None
        check(lib.rrl_line_tri_scan_ex(_p(line), ws, nb, B, N, M, L, _MODES[mode], int(chunk), op, s),
              "rrl_line_tri_scan")
        check(lib.rrl_line_pair_dist_ex(_p(tri1), _p(tri2), _p(line), ws, nb, B, N, M, L, s_m, s_n, e_m,
                                        e_n, int(pool), op, s), "rrl_line_pair_dist")
        check(lib.rrl_loss_reduce_ex(ws, nb, _p(st.loss), B, N, M, L, s_m, s_n, e_m, e_n, int(pool), op, s),
              "rrl_loss_reduce")
    _keep_ride(st, ride)
    return st


class _IntersectionLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points1, points2, line, rng, pool, mode, chunk, target_from=None, opts=None):
        dev = _home(points1, points2, line)
        tri1, tri2 = _prep(points1, "points1", 9, dev), _prep(points2, "points2", 9, dev)
        ln = _prep(line, "line", 6, dev)
        if tri1.dim() != 3 or tri2.dim() != 3 or ln.dim() != 3:
            raise ValueError("Input is wrong: points1/points2/line must be 3-D (B, n, c)")
        if not (tri1.shape[0] == tri2.shape[0] == ln.shape[0]):
            raise ValueError("points1, points2 and line must share the batch dimension")
        if tri1.shape[0] == 0 or ln.shape[1] == 0:  # empty batch / no lines: nothing to launch
            ctx.st = None
            ctx.set_materialize_grads(False)
            dev, G = tri1.device, (1 if pool else tri1.shape[0])
            out = (torch.zeros(G, device=dev), torch.zeros(G, 4, dtype=torch.int32, device=dev),
                   torch.zeros(4, dtype=torch.int32, device=dev))
            ctx.mark_non_differentiable(out[1], out[2])
            return out
        st = loss_forward_raw(tri1, tri2, ln, rng, pool, mode, chunk, target_from=target_from, opts=opts)
        ctx.st, ctx.tri1, ctx.tri2, ctx.pool = st, tri1, tri2, bool(pool)
        ctx.in_devs = (points1.device, points2.device)
        info, status = st.info, st.status
        ctx.mark_non_differentiable(info, status)
        ctx.set_materialize_grads(False)  # no zero-filled grads for the integer outputs
        _IntersectionLoss.last_state = st  # for shard_payload(): the newest evaluation
        # return a view: st.loss itself must not become the autograd output, or last_state would
        # keep the graph (and the AccumulateGrad nodes of the inputs) alive across iterations
        return st.loss.view(-1), info, status

    @staticmethod
    def backward(ctx, g_loss, _g1, _g2):
        if g_loss is None or ctx.st is None:
            return (None,) * 9
        lib = _lib.load()
        st, tri1, tri2 = ctx.st, ctx.tri1, ctx.tri2
        B, N, _ = tri1.shape
        M, L = tri2.shape[1], st.dims[3]
        g = g_loss.detach().to(device=tri1.device, dtype=torch.float32).contiguous()
        g1 = torch.empty_like(tri1)  # zeroed by rrl_loss_backward
        g2 = torch.empty_like(tri2) if ctx.needs_input_grad[1] else None
        with _guard(tri1.device):
            check(lib.rrl_loss_backward(_p(tri1), _p(tri2), _p(st.ws), st.nbytes, _p(g), _p(g1), _p(g2),
                                        B, N, M, L, int(ctx.pool), _stream(tri1.device)), "rrl_loss_backward")
        g1 = g1.to(ctx.in_devs[0]) if ctx.needs_input_grad[0] else None
        if g2 is not None:
            g2 = g2.to(ctx.in_devs[1])
        return g1, g2, None, None, None, None, None, None, None


# ---- the reference-signature call (loss.cal_loss_intersection_batch_whole_median_pts_lines) ----------------------
# Every reference trainer calls it once per SAMPLE inside a Python loop and tests / adds the result
# (rpm/Train_RPM.py:226-231, dcp/Train_DCP.py:266-270, fmr/model.py:302-306), so its cost is host time: workspaces
# are leased from a per-shape pool instead of allocated, the forward and its one read-back are a single C call
# (rrl_loss_forward_info: launches + 16-byte copy into pinned memory + wait), and nothing but the loss tensor
# escapes (a fresh tensor per call: a pooled buffer would alias the results of successive calls).
_pool = {}          # (B, N, M, L, G, device index, raw stream) -> [LossState]
_pool_bytes = [0]
_POOL_CAP = 2 << 30


def _lease_state(B, N, M, L, G, dev):
    """A LossState nobody else refers to: from the pool of its shape / device / stream when one is free (no live
    autograd node, not the `last_state`), else a new one (which joins the pool).  Reuse is stream-ordered like the
    caching allocator's; never pooled while a graph is being captured (capture-time allocations belong to the
    graph's private pool)."""
    if torch.cuda.is_current_stream_capturing():
        return LossState(B, N, M, L, G, dev)
    key = (B, N, M, L, G, dev.index, _raw_stream(dev.index))
    lst = _pool.get(key)
    if lst is None:
        lst = _pool[key] = []
    for st in lst:
        if sys.getrefcount(st) == 3:  # the pool's list, this loop variable, getrefcount's own argument
            st.target_state = None
            return st
    st = LossState(B, N, M, L, G, dev)
    st.wsp = _p(st.ws)
    st.host_info = torch.empty(4 * G, dtype=torch.int32).pin_memory()
    st.hostp = ctypes.c_void_p(st.host_info.data_ptr())
    if _pool_bytes[0] + st.nbytes > _POOL_CAP:  # many distinct shapes (ragged data): start over
        _pool.clear()
        _pool_bytes[0] = 0
        lst = _pool[key] = []
    if len(lst) < 64:
        lst.append(st)
        _pool_bytes[0] += st.nbytes
    return st


def _dropin_forward(ctx, points1, points2, line, rng, pool, mode, chunk):
    """Forward of the reference-signature call: rrl_loss_forward_info on a leased workspace (one C call incl. the read-back
    of the flags).  Returns loss (G,) -- a fresh tensor -- and leaves the flags of all groups in _DropinLoss.flags_all."""
    dev = _home(points1, points2, line)
    tri1, tri2 = _prep(points1, "points1", 9, dev), _prep(points2, "points2", 9, dev)
    ln = _prep(line, "line", 6, dev)
    if tri1.dim() != 3 or tri2.dim() != 3 or ln.dim() != 3:
        raise ValueError("Input is wrong: points1/points2/line must be 3-D (B, n, c)")
    B, N, _ = tri1.shape
    M, L = tri2.shape[1], ln.shape[1]
    if not (tri2.shape[0] == ln.shape[0] == B):
        raise ValueError("points1, points2 and line must share the batch dimension")
    G = 1 if pool else B
    ctx.set_materialize_grads(False)
    if B == 0 or L == 0:  # empty batch / no lines: nothing to launch, no bucket
        ctx.st = None
        _DropinLoss.flags_all = [0, 0, 0, 0] * max(G, 1)
        _DropinLoss.flags = [0, 0, 0, 0]
        return torch.zeros(max(G, 1), device=dev)
    s_m, s_n, e_m, e_n = _check_range(rng)
    st = _lease_state(B, N, M, L, G, dev)
    if not hasattr(st, "hostp"):  # created while capturing: no read-back possible there
        raise RRLError("the reference-signature loss synchronises (it returns None / raises on the host): "
                       "it cannot be captured into a graph; use batched_intersection_loss / ops.registration_loss")
    st.loss = loss = torch.empty(G, dtype=torch.float32, device=dev)
    with _guard(dev):
        check(_lib.load().rrl_loss_forward_info(_p(tri1), _p(tri2), _p(ln), st.wsp, st.nbytes, _p(loss), B, N, M, L,
                                                s_m, s_n, e_m, e_n, int(pool), _MODES[mode], int(chunk), None,
                                                st.hostp, _stream(dev)), "rrl_loss_forward_info")
    _DropinLoss.flags_all = fl = st.host_info.tolist()  # 4 per group
    _DropinLoss.flags = fl[:4]
    ctx.st, ctx.tri1, ctx.tri2, ctx.pool = st, tri1, tri2, bool(pool)
    ctx.in_devs = (points1.device, points2.device)
    _IntersectionLoss.last_state = st
    return loss


def _dropin_backward(ctx, g_loss):
    """(grad points1, grad points2) of a _dropin_forward evaluation for dL/dloss = g_loss (G,)."""
    st, tri1, tri2 = ctx.st, ctx.tri1, ctx.tri2
    B, N, M, L, _ = st.dims
    dev = tri1.device
    g = g_loss if (g_loss.device == dev and g_loss.dtype == torch.float32 and g_loss.is_contiguous()) else \
        g_loss.detach().to(device=dev, dtype=torch.float32).contiguous()
    g1 = torch.empty_like(tri1)  # zeroed by rrl_loss_backward
    g2 = torch.empty_like(tri2) if ctx.needs_input_grad[1] else None
    with _guard(dev):
        check(_lib.load().rrl_loss_backward(_p(tri1), _p(tri2), st.wsp, st.nbytes, _p(g), _p(g1), _p(g2),
                                            B, N, M, L, int(ctx.pool), _stream(dev)), "rrl_loss_backward")
    if not ctx.needs_input_grad[0]:
        g1 = None
    elif ctx.in_devs[0] != dev:
        g1 = g1.to(ctx.in_devs[0])
    if g2 is not None and ctx.in_devs[1] != dev:
        g2 = g2.to(ctx.in_devs[1])
    return g1, g2


class _DropinLoss(torch.autograd.Function):
    """forward = rrl_loss_forward_info on a leased workspace; the host flags of the call are left in
    _DropinLoss.flags ([nbuckets, nselected, nvalues, NaN flag] of group 0; flags_all: of every group)."""
    flags = None
    flags_all = None

    @staticmethod
    def forward(ctx, points1, points2, line, rng, pool, mode, chunk):
        return _dropin_forward(ctx, points1, points2, line, rng, pool, mode, chunk).view(-1)  # a view: st.loss itself
        #                                      must not become the autograd output (st -> loss -> node -> st)

    @staticmethod
    def backward(ctx, g_loss):
        if g_loss is None or ctx.st is None:
            return (None,) * 7
        return (*_dropin_backward(ctx, g_loss), None, None, None, None, None)


class _DropinBatch(torch.autograd.Function):
    """The whole batch of a trainer's per-sample loop as ONE node with B outputs of shape (1,): the caller's
    `total = total + one_j ... ; total.backward()` hands this node B gradients at once -- no slice / scatter kernels of
    autograd's own, one rrl_loss_backward for the batch."""

    @staticmethod
    def forward(ctx, points1, points2, line, rng, mode, chunk):
        loss = _dropin_forward(ctx, points1, points2, line, rng, False, mode, chunk)
        # B one-element tensors on the batch's loss buffer that autograd does NOT know as views of each other (set_ on the
        # shared storage): a caller may modify a returned loss in place (`total = first; total += second`), which outputs
        # that are views of one tensor would refuse ("function that returns multiple views")
        st_, off = loss.untyped_storage(), loss.storage_offset()
        return tuple(torch.empty(0, dtype=loss.dtype, device=loss.device).set_(st_, off + j, (1,), (1,)) for j in range(loss.numel()))

    @staticmethod
    def backward(ctx, *grads):
        if ctx.st is None or all(g is None for g in grads):
            return (None,) * 6
        zero = None
        parts = []
        for g in grads:
            if g is None:
                if zero is None:
                    zero = ctx.tri1.new_zeros(1)
                g = zero
            parts.append(g.reshape(1))
        return (*_dropin_backward(ctx, torch.cat(parts)), None, None, None, None)


# ---- the trainers' literal loop, served by ONE evaluation of the whole batch (round 4) ----------------------------
# `for j in range(B): loss += cal_loss...(1, 1, 5, 5, p1[j:j+1], p2[j:j+1], line[j:j+1])` (rpm/Train_RPM.py:226-231,
# dcp/Train_DCP.py:266-270, fmr/model.py:302-306) hands this module B slices of three batch tensors.  A slice knows its
# base (Tensor._base) and its offset in it, so the FIRST call of such a loop evaluates all B samples at once with the
# independent-sample op (bit-identical per sample to B single calls, test_batched_equals_per_sample), reads all B flag
# rows back in its one host sync, and the calls for the other j are served from that evaluation: each returned loss is a
# slice of ONE autograd node, so the caller's `total.backward()` runs one batched backward.  A hit needs the same base
# OBJECTS (held weakly), unchanged version counters of all three, equal shapes / bucket range / mode and the same grad
# mode; anything else -- and a batch whose scan saw a NaN, which the per-call path attributes to its sample -- takes the
# per-call path.  RRL_DROPIN_BATCH=0 (or ops.DROPIN_BATCH = False) turns it off; RRL_DROPIN_BATCH_MAX bounds B (64).
DROPIN_BATCH = os.environ.get("RRL_DROPIN_BATCH", "1") != "0"
DROPIN_BATCH_MAX = int(os.environ.get("RRL_DROPIN_BATCH_MAX", "64"))
_batch_cache = [None]
dropin_batch_stats = {"evaluations": 0, "served": 0, "nan_fallbacks": 0}


class _BatchEval:
    __slots__ = ("bases", "key", "loss", "flags", "nan", "geo")


def _slice_of_batch(t, last):
    """(base, B, j, n) when t == base.view(B, n, last)[j:j+1] for a contiguous base with B >= 2, else None."""
    b = t._base
    if b is None or t.dim() != 3 or t.shape[0] != 1 or t.shape[2] != last or not t.is_cuda or t.dtype != torch.float32:
        return None
    if not (t.is_contiguous() and b.is_contiguous()):
        return None
    per = t.shape[1] * last
    tot = b.numel()
    if per == 0 or tot % per or tot // per < 2:
        return None
    off = t.storage_offset() - b.storage_offset()
    if off < 0 or off % per or off // per >= tot // per:
        return None
    return b, tot // per, off // per, t.shape[1]


def dropin_batch_clear():
    """Drop the cached whole-batch evaluation (it keeps one workspace and the batch's autograd node alive)."""
    _batch_cache[0] = None


def _serve_from_batch(points1, points2, line, rng, mode, chunk):
    c = _batch_cache[0]
    if not INTERNALS["tensor_base"]:  # (a torch without Tensor._base: the per-call path serves the loop)
        _warn_once("Tensor._base")
        return None
    b1, b2, bl = points1._base, points2._base, line._base
    if b1 is None or b2 is None or bl is None:
        return None
    grad = torch.is_grad_enabled() and (b1.requires_grad or b2.requires_grad)
    if c is not None and c.bases[0]() is b1 and c.bases[1]() is b2 and c.bases[2]() is bl:
        # the usual call of a loop: same bases as the cached evaluation -- is it still valid, and which sample is this?
        B, N, M, L, o1, o2, ol = c.geo
        if c.key == (_write_key(b1), _write_key(b2), _write_key(bl), rng, mode, chunk, grad) and points1.shape == (1, N, 9) \
                and points2.shape == (1, M, 9) and line.shape == (1, L, 6) and points1.is_contiguous() \
                and points2.is_contiguous() and line.is_contiguous():
            j, r = divmod(points1.storage_offset() - o1, N * 9)
            if r == 0 and 0 <= j < B and points2.storage_offset() - o2 == j * M * 9 and line.storage_offset() - ol == j * L * 6:
                if c.nan:
                    return None
                dropin_batch_stats["served"] += 1
                return c.loss[j], c.flags[4 * j:4 * j + 4]
    import weakref
    s1, s2, sl = _slice_of_batch(points1, 9), _slice_of_batch(points2, 9), _slice_of_batch(line, 6)
    if s1 is None or s2 is None or sl is None:
        return None
    (b1, B, j, N), (b2, B2, j2, M), (bl, BL, jl, L) = s1, s2, sl
    if not (B == B2 == BL and j == j2 == jl) or B > DROPIN_BATCH_MAX or not (b1.device == b2.device == bl.device):
        return None
    c = _BatchEval()
    c.bases = (weakref.ref(b1), weakref.ref(b2), weakref.ref(bl))
    c.key = (_write_key(b1), _write_key(b2), _write_key(bl), rng, mode, chunk, grad)
    c.geo = (B, N, M, L, b1.storage_offset(), b2.storage_offset(), bl.storage_offset())
    c.loss = _DropinBatch.apply(b1.view(B, N, 9), b2.view(B, M, 9), bl.view(B, L, 6), rng, mode, chunk)  # B outputs
    c.flags = _DropinLoss.flags_all
    c.nan = bool(c.flags[3])  # one flag per launch: which sample it was is the per-call path's to say
    if c.nan:
        c.loss = None
        dropin_batch_stats["nan_fallbacks"] += 1
    dropin_batch_stats["evaluations"] += 1
    _batch_cache[0] = c
    if c.nan:
        return None
    dropin_batch_stats["served"] += 1
    return c.loss[j], c.flags[4 * j:4 * j + 4]


def intersection_loss_dropin(points1, points2, line, rng=(1, 1, 5, 5), pool=False, mode="cull", chunk=0):
    """(loss (G,) with a grad_fn, [nbuckets, nselected, nvalues, NaN flag] as Python ints) -- the forward of the
    reference-signature call with its single host read-back inside (one C call).  See _DropinLoss; a call whose three
    arguments are the [j:j+1] slices of batch tensors is served from one evaluation of the whole batch (above)."""
    rng = tuple(rng)
    if DROPIN_BATCH and isinstance(points1, torch.Tensor) and points1.dim() == 3 and points1.shape[0] == 1 \
            and isinstance(points2, torch.Tensor) and isinstance(line, torch.Tensor):
        hit = _serve_from_batch(points1, points2, line, rng, mode, chunk)
        if hit is not None:
            return hit
    loss = _DropinLoss.apply(points1, points2, line, rng, pool, mode, chunk)
    return loss, _DropinLoss.flags


def _with_ride(opts, order1, order2, chamfer, B, N, M, dev, problems=0):
    """opts for a call that takes order1= / order2= / chamfer=True as keywords (or a ready make_opts object).
    problems = Bt > 0: a multi-pose evaluation of B instances (include/rrl.h rrl_opts.problems); the orders then have Bt rows."""
    if opts is not None:
        if chamfer and getattr(opts, "ride", None) is None:
            raise ValueError("chamfer=True with ready-made opts: build them with make_opts(chamfer=ChamferRide(...))")
        return opts
    ride = ChamferRide(B, N, M, dev) if chamfer else None
    if ride is None and order1 is None and order2 is None and not problems:
        return None
    # the kernels read an order as int32 [B][64 ceil(n / 64)] on the op's GPU: anything else would be read out of bounds
    Bo = problems or B
    return make_opts(order1=_check_order(order1, Bo, N, dev, "order1"), order2=_check_order(order2, Bo, M, dev, "order2"),
                     chamfer=ride, problems=problems)


def intersection_loss(points1, points2, line, rng=(1, 1, 5, 5), pool=False, mode="cull", chunk=0,
                      target_from=None, order1=None, order2=None, opts=None, chamfer=False):
    """Batched loss: returns (loss[G], info[G,4] = (nbuckets, nselected, nvalues, NaN flag), status[4])
    on the GPU, G = 1 if pool else B.  Each sample is an independent loss (what every reference
    caller obtains by looping B=1 calls); pool=True reproduces the reference's own B>1 behaviour
    (SURVEY Q2).  No host synchronisation happens here.
    order1 / order2: ops.cloud_order of the two clouds (in any rigid pose of them) -- the per-call cell sort is skipped,
    same results; opts: make_opts(...) for anything else per call.
    chamfer=True: the Chamfer monitor between the clouds' first points (chamfer_from_state) is issued inside this
    evaluation's scan launch (ChamferRide); chamfer_from_state(last_state()) then returns it without a launch."""
    if opts is None and (chamfer or order1 is not None or order2 is not None):
        opts = _with_ride(None, order1, order2, chamfer and not pool, points1.shape[0], points1.shape[1], points2.shape[1],
                          _home(points1, points2, line))
    return _IntersectionLoss.apply(points1, points2, line, tuple(rng), pool, mode, chunk, target_from, opts)


def shard_payload(loss, gR=None, gt=None, state=None):
    """[sum of valid losses, #valid, sum_b gR (9), sum_b gt (3)] as one (14,) tensor in one
    launch -- the buffer a rank contributes to the all-reduce (rrl_hip.dist).  `state` defaults
    to the LossState of the latest intersection_loss call."""
    st = state or _IntersectionLoss.last_state
    B, N, M, L, G = st.dims
    dev = st.ws.device
    out = torch.empty(14, dtype=torch.float32, device=dev)
    gRc = _prep(gR, "gR", None, dev) if gR is not None else None
    gtc = _prep(gt, "gt", None, dev) if gt is not None else None
    with _guard(dev):
        check(_lib.load().rrl_shard_payload(_p(_prep(loss, "loss", None, dev)), _p(st.ws), st.nbytes, _p(gRc),
                                            _p(gtc), _p(out), G, N, M, L, _stream(dev)), "rrl_shard_payload")
    return out


class _RegistrationLoss(torch.autograd.Function):
    """Fused training op: rigid transform of the source pseudo-triangles + loss (one C call
    each way; the transformed triangles and their gradient live in the workspace)."""

    @staticmethod
    def forward(ctx, src_tri, R, t, tar_tri, line, rng, transpose_r, mode, chunk, want_payload,
                target_from=None, opts=None):
        dev = _home(src_tri, tar_tri, line, R, t)
        src = _prep(src_tri, "src_tri", 9, dev)
        tri2, ln = _prep(tar_tri, "tar_tri", 9, dev), _prep(line, "line", 6, dev)
        Rm, tv = _prep(R, "R", None, dev).reshape(-1, 3, 3), _prep(t, "t", None, dev).reshape(-1, 3)
        if src.dim() != 3 or tri2.dim() != 3 or ln.dim() != 3:
            raise ValueError("src_tri/tar_tri/line must be 3-D (B, n, c)")
        Bt, N, _ = src.shape
        M, L = tri2.shape[1], ln.shape[1]
        B = Rm.shape[0]  # instances: B == Bt, or k poses of each of the Bt problems (multi-pose, rrl_opts.problems)
        if not (tri2.shape[0] == ln.shape[0] == Bt and tv.shape[0] == B) or (B != Bt and (Bt == 0 or B % Bt)):
            raise ValueError("batch dimensions differ (src_tri / tar_tri / line share B_t; R, t hold B_t or k * B_t poses)")
        multi = B != Bt
        if multi and (opts is None or getattr(opts, "problems", 0) != Bt):
            raise ValueError("multi-pose call without rrl_opts.problems (use ops.registration_loss, which sets it)")
        if B == 0 or L == 0:  # empty batch / no lines: nothing to launch, zero loss, zero gradient
            ctx.st = None
            ctx.set_materialize_grads(False)
            dev = src.device
            out = (torch.zeros(B, device=dev), torch.zeros(B, 4, dtype=torch.int32, device=dev),
                   torch.zeros(4, dtype=torch.int32, device=dev))
            _IntersectionLoss.last_state = None
            ctx.mark_non_differentiable(out[1], out[2])
            return out
        s_m, s_n, e_m, e_n = _check_range(rng)
        st = LossState(B, N, M, L, B, src.device)
        ride = _arm_ride(opts)
        with _guard(dev):
            check(_lib.load().rrl_registration_forward_ex(
                _p(src), _p(Rm), _p(tv), _p(tri2), _p(ln), _p(st.ws), st.nbytes, _p(st.loss), B, N, M, L,
                int(transpose_r), s_m, s_n, e_m, e_n, _MODES[mode], int(chunk),
                _target_ws(target_from, B, N, M, L), _optr(opts), _stream(dev)), "rrl_registration_forward")
        st.target_state = getattr(target_from, "target_state", None) or target_from  # whose workspace holds cloud 2
        _keep_ride(st, ride)
        ctx.st, ctx.src, ctx.Rm, ctx.tri2 = st, src, Rm, tri2
        ctx.meta = (int(transpose_r), bool(want_payload), R.shape, t.shape, src_tri.device, R.device, t.device)
        ctx.opts = opts
        ctx.multi = multi
        info, status = st.info, st.status
        ctx.mark_non_differentiable(info, status)
        ctx.set_materialize_grads(False)
        _IntersectionLoss.last_state = st
        return st.loss.view(-1), info, status  # a view: see _IntersectionLoss.forward

    @staticmethod
    def backward(ctx, g_loss, _g1, _g2):
        if g_loss is None or ctx.st is None:
            return (None,) * 12
        st, src, Rm, tri2 = ctx.st, ctx.src, ctx.Rm, ctx.tri2
        tr, want_payload, Rshape, tshape, sdev, Rdev, tdev = ctx.meta
        B, N, M, L, _ = st.dims
        if ctx.multi and ctx.needs_input_grad[0]:
            raise RRLError("a multi-pose evaluation returns dL/dR, dL/dt only (src_tri must not require grad)")
        g = g_loss if (g_loss.device == src.device and g_loss.is_contiguous()) else \
            g_loss.to(device=src.device, dtype=torch.float32).contiguous()
        if not ctx.needs_input_grad[0] and not getattr(st, "gacc_used", False):
            out = st.gacc  # zeroed by the forward; the direct backward accumulates into it
            st.gacc_used = True
        else:  # a second backward over the same forward, or the d/dsrc route (which overwrites)
            out = torch.empty(B * 12 + 14, dtype=torch.float32, device=src.device)
        gR, gt, payload = out[:B * 9], out[B * 9:B * 12], out[B * 12:B * 12 + 14]
        gsrc = torch.empty_like(src) if ctx.needs_input_grad[0] else None
        with _guard(src.device):
            check(_lib.load().rrl_registration_backward_ex(
                _p(src), _p(Rm), _p(tri2), _p(st.ws), st.nbytes, _p(st.loss), _p(g), _p(gsrc), _p(gR),
                _p(gt), _p(payload) if want_payload else None, B, N, M, L, tr, _optr(ctx.opts), _stream(src.device)),
                "rrl_registration_backward")
        st.payload = payload if want_payload else None
        return (gsrc.to(sdev) if gsrc is not None else None, gR.reshape(Rshape).to(Rdev),
                gt.reshape(tshape).to(tdev), None, None, None, None, None, None, None, None, None)


def registration_loss(src_tri, R, t, tar_tri, line, rng=(1, 1, 5, 5), transpose_r=True,
                      mode="cull", chunk=0, want_payload=False, target_from=None, order1=None, order2=None, opts=None,
                      chamfer=False):
    """loss[b] of `src_tri[b]` moved by (R[b], t[b]) against `tar_tri[b]` along `line[b]` -- the
    rigid transform of the training call sites fused with the loss.  transpose_r=True is
    x R^T + t (R x + t per point: RPM / DCP / FMR), False is x R + t (Reconstruction_point).
    Returns (loss (B,), info (B,4), status (4,)); differentiable in R, t and src_tri.
    want_payload=True also builds the 14-float batch-shard payload during backward
    (LossState.payload of the call, see rrl_hip.dist).
    target_from: LossState (ops.last_state()) of an earlier call with the SAME tar_tri and line:
    the target cloud is not scanned again (RPM / FMR: several poses, one target, one line set).
    order1 / order2: ops.cloud_order of src_tri (in its own frame: a rigid motion keeps the order) and tar_tri -- the
    per-call cell sort is skipped, same results.
    chamfer=True: the trainers' Chamfer monitor (moved source's first points vs the target's; chamfer_from_state) is issued
    inside this evaluation's scan launch (ChamferRide) -- chamfer_from_state(last_state()) then costs no launch."""
    # MULTI-POSE (round 5; include/rrl.h rrl_opts.problems): R / t may hold k poses for each of the B_t problems
    # (instance s = pose s // B_t of problem s % B_t -- torch.cat of the per-iteration (B_t, 3, 3) estimates): all k * B_t
    # losses come out of ONE evaluation, the target scanned once per problem; loss (k * B_t,), bit-identical per instance to
    # k separate calls.  (What RPM's num_iter and FMR's last three estimates evaluate: callsites.)
    Bt = src_tri.shape[0]
    Binst = R.reshape(-1, 3, 3).shape[0]
    problems = Bt if (Binst != Bt and Bt > 0 and Binst % Bt == 0) else 0
    if problems and (target_from is not None or mode != "cull"):
        raise ValueError("a multi-pose evaluation scans its target itself, in scan mode cull")
    if opts is None and (chamfer or order1 is not None or order2 is not None or problems):
        opts = _with_ride(None, order1, order2, chamfer, Binst, src_tri.reshape(Bt, -1, 9).shape[1],
                          tar_tri.reshape(tar_tri.shape[0], -1, 9).shape[1], _home(src_tri, tar_tri, line, R, t), problems)
    return _RegistrationLoss.apply(src_tri, R, t, tar_tri, line, tuple(rng), transpose_r, mode,
                                   chunk, want_payload, target_from, opts)


class RegistrationStep:
    """The fused training op WITHOUT autograd and without a graph: forward + backward to (dR, dt) as ONE C call
    (rrl_registration_step: where the tail kernel serves the shape the backward rides in the reduce's launch, 5 launches
    per step) on buffers allocated once -- what a training loop needs when (R, t) come out of a network:

        step = ops.RegistrationStep(src_tri, tar_tri, L)            # once per shape
        loss, gR, gt = step(R.detach(), t.detach(), lines)[:3]      # every iteration (gout = ones)
        torch.autograd.backward([R, t], [gR, gt])                   # hand the gradient to the network

    Issued this way the C2 step is GPU-bound (about 30 us of host time per step); replaying the
    launches as a hipGraph costs ~6 us more (a replay has ~8 us of fixed cost + 1.5 us per node on this
    stack, tools/attic/graph_node_cost.py), and the autograd front end (registration_loss) is host-bound when
    issued eagerly.  Same kernels, same numbers as registration_loss.  The outputs are views of buffers
    that the next call overwrites.  want_payload: also the 14-float batch-shard payload (rrl_hip.dist).

    prepared=True (default; round 4): the spatial ORDER of both clouds is computed once (ops.cloud_order, at
    construction and whenever a new cloud arrives through src_tri= / tar_tri= without its src_order= / tar_order=) and
    every step runs the prepared build -- one wide launch moves the source, writes the records at their sorted
    positions and refits the sphere tree; the cell sort (10 us of a 70 us step) is gone -- and a target that has not
    changed since the previous call (same tensor, same torch version counter) is not rebuilt at all
    (RRL_F_TARGET_KEPT).  Bit-identical loss either way; prepared=False is the cold path (records + sort each step), the
    right choice when every step brings new clouds that are evaluated once."""

    # rrl_registration_step; False (or RRL_ONE_CALL=0): rrl_registration_forward_cached + rrl_registration_backward
    ONE_CALL = os.environ.get("RRL_ONE_CALL", "1") != "0"

    def __init__(self, src_tri, tar_tri, n_lines, rng=(1, 1, 5, 5), transpose_r=True, mode="cull", chunk=0,
                 want_payload=False, prepared=None, src_order=None, tar_order=None, reduce_mode=None, deterministic=None,
                 sort_parts=None, chamfer=False, poses=1, chain=False):
        """reduce_mode / deterministic / sort_parts: per-call options of THIS step object (include/rrl.h rrl_opts; None =
        the library default) -- they travel with every call, so two steps with different options can run from two
        threads on two streams at the same time (tests/test_gpu_threads.py).
        poses = k > 1 (round 5): a MULTI-POSE step (rrl_opts.problems) -- every call takes R (k * B, 3, 3), t (k * B, 3): k
        poses of each of the B problems (instance i * B + b = pose i of problem b; RPM's num_iter, FMR's last estimates),
        evaluated in ONE set of launches with the target scanned once per problem; loss (k * B,), gR (k * B, 3, 3),
        gt (k * B, 3), bit-identical per instance to k single-pose steps.  Scan mode cull, clouds <= 65536 triangles."""
        dev = _home(src_tri, tar_tri)
        self.dev = dev
        self._extra = dict(reduce_mode=reduce_mode, deterministic=deterministic, sort_parts=sort_parts)
        self._want_chamfer = bool(chamfer)  # the Chamfer monitor of every step, issued inside its scan launch (ChamferRide)
        self.ride = self.chamfer_value = None
        self.src = _prep(src_tri, "src_tri", 9, dev)
        self.tar = _prep(tar_tri, "tar_tri", 9, dev)
        if self.src.dim() != 3 or self.tar.dim() != 3 or self.src.shape[0] != self.tar.shape[0]:
            raise ValueError("src_tri/tar_tri must be (B, n, 9) with the same B")
        Bt, N, _ = self.src.shape
        M, L = self.tar.shape[1], int(n_lines)
        if Bt == 0 or L <= 0:
            raise ValueError("RegistrationStep needs a non-empty batch and line set")
        self.poses = int(poses)
        if self.poses < 1 or (self.poses > 1 and (mode != "cull" or max(N, M) > 65536)):
            raise ValueError("poses must be >= 1; a multi-pose step runs scan mode cull on clouds <= 65536 triangles")
        B = Bt * self.poses  # instances of one call
        self.Bt = Bt
        if self.poses > 1:
            self._extra["problems"] = Bt
        self.dims = (B, N, M, L)
        self.rng = _check_range(rng)
        self.tr, self.mode, self.chunk = int(bool(transpose_r)), _MODES[mode], int(chunk)
        self.st = LossState(B, N, M, L, B, dev)
        if self._want_chamfer:
            self.ride = ChamferRide(B, N, M, dev)
            self._extra["chamfer"] = self.ride
        self.ones = torch.ones(B, dtype=torch.float32, device=dev)
        gacc = self.st.gacc
        self.gR, self.gt = gacc[:B * 9].view(B, 3, 3), gacc[B * 9:B * 12].view(B, 3)
        self.payload = gacc[B * 12:B * 12 + 14] if want_payload else None
        self._lib = _lib.load()
        self._fixed_f = (_p(self.st.ws), self.st.nbytes, _p(self.st.loss), B, N, M, L, self.tr, *self.rng, self.mode,
                         self.chunk, None)
        self._fixed_b = (_p(self.st.ws), self.st.nbytes, _p(self.st.loss))
        self._tail_b = (None, _p(gacc[:B * 9]), _p(gacc[B * 9:B * 12]), _p(self.payload), B, N, M, L, self.tr)
        self._head_s = (_p(self.st.ws), self.st.nbytes, _p(self.st.loss))
        self._tail_s = (_p(gacc[:B * 9]), _p(gacc[B * 9:B * 12]), _p(self.payload), B, N, M, L, self.tr, *self.rng,
                        self.mode, self.chunk, None)
        if prepared is None:  # default: on (RRL_PREPARED=0 turns the default off: A/B runs of unmodified callers)
            prepared = os.environ.get("RRL_PREPARED", "1") != "0"
        self.prepared = bool(prepared) and mode == "cull" and max(N, M) <= 65536
        self._out = (self.st.loss.view(-1), self.gR, self.gt, self.payload, self.st.info)  # (static views: built once)
        self._kept_key = None  # _write_key of the target whose records the workspace holds
        self.keep_target = True  # False: rebuild the target's records in every call (see invalidate_target)
        # chained steps (round 6; include/rrl.h RRL_F_CHAIN / RRL_F_CHAINED; see LossStep).  OFF by default here: a chained
        # step clears its hit counts on exit, and this step's state is what callers hand on as target_from= (RPM / FMR)
        self.chain = bool(chain)
        self._chain_left = ctypes.c_int32(0)
        self._chain_ready = False
        self.fused = False
        self.order1 = self.order2 = None
        if self.prepared:
            self.order1 = _check_order(src_order, Bt, N, dev, "src_order") if src_order is not None else cloud_order(self.src)
            self.order2 = _check_order(tar_order, Bt, M, dev, "tar_order") if tar_order is not None else cloud_order(self.tar)
        self._set_opts()

    def invalidate_target(self):
        """Forget the kept target: the next call rebuilds cloud 2's records and tree.  The step notices a changed target by
        itself through (data_ptr, torch's version counter, the library's own raw-write serial); what it cannot see is a
        write that bypasses both -- `tar.data.copy_()`, a custom kernel or another raw-pointer library writing into the
        same buffer -- : call this after such a write, or set `.keep_target = False` to rebuild in every call.  (A hipGraph
        capture bakes in whichever variant is active at capture time: capture with keep_target = False when the target's
        CONTENT changes between replays.)"""
        self._kept_key = None

    def _set_opts(self):
        self._opts = make_opts(order1=self.order1, order2=self.order2, **self._extra)
        self._opts_kept = make_opts(order1=self.order1, order2=self.order2, target_kept=True, **self._extra) if self.prepared else self._opts
        self._optr = ctypes.byref(self._opts) if self._opts is not None else None
        self._optr_kept = ctypes.byref(self._opts_kept) if self._opts_kept is not None else None
        self._chain_ready = False
        if self.prepared:  # first / kept / chained (as LossStep)
            kw = dict(order1=self.order1, order2=self.order2, chain_left=self._chain_left, **self._extra)
            self._opts_c = (make_opts(chain=_lib.F_CHAIN, **kw), make_opts(target_kept=True, chain=_lib.F_CHAIN, **kw),
                            make_opts(target_kept=True, chain=_lib.F_CHAIN | _lib.F_CHAINED, **kw))
            self._optr_c = tuple(ctypes.byref(o) for o in self._opts_c)

    def __call__(self, R, t, line, grad_loss=None, src_tri=None, tar_tri=None, target_from=None, src_order=None,
                 tar_order=None):
        """src_tri / tar_tri: this step's clouds (same shapes as at construction); default: the tensors given
        to the constructor (update those in place, or pass the new batch here).
        target_from: the LossState (another step's .st, or ops.last_state()) of an evaluation with the SAME tar_tri and
        line -- RPM / FMR evaluate several poses against one target and one line set --: only the source is prepared,
        sorted and scanned here, the target's hit lists are taken from that state (bit-identical results)."""
        B, N, M, L = self.dims
        Bt = self.Bt  # problems: the clouds, orders and lines have Bt entries; B = poses * Bt instances
        dev = self.dev
        if src_tri is not None:
            self.src = _prep(src_tri, "src_tri", 9, dev)
        if tar_tri is not None:
            self.tar = _prep(tar_tri, "tar_tri", 9, dev)
        if tuple(self.src.shape) != (Bt, N, 9) or tuple(self.tar.shape) != (Bt, M, 9):
            raise ValueError(f"src_tri {(Bt, N, 9)} / tar_tri {(Bt, M, 9)} expected")
        if self.poses > 1 and target_from is not None:
            raise ValueError("a multi-pose step scans its target itself")
        op = self._optr
        if self.prepared:
            if src_tri is not None or src_order is not None or tar_tri is not None or tar_order is not None:
                if src_tri is not None or src_order is not None:  # a new source: its order comes along, or is taken now
                    self.order1 = _check_order(src_order, Bt, N, dev, "src_order") if src_order is not None else cloud_order(self.src)
                if tar_tri is not None or tar_order is not None:
                    self.order2 = _check_order(tar_order, Bt, M, dev, "tar_order") if tar_order is not None else cloud_order(self.tar)
                    self._kept_key = None
                self._set_opts()
            key = _write_key(self.tar) if (self.keep_target and target_from is None) else None
            kept = key is not None and key == self._kept_key
            if self.chain and RegistrationStep.ONE_CALL and target_from is None:
                op = self._optr_c[2 if (kept and self._chain_ready) else (1 if kept else 0)]
            elif kept:
                op = self._optr_kept
            self._kept_key = None  # (set again below, once the call has been issued: a call that raises keeps nothing)
        self._chain_ready = False
        self._chain_left.value = 0
        Rm, tv, ln = _prep(R, "R", None, dev), _prep(t, "t", None, dev), _prep(line, "line", 6, dev)
        if Rm.numel() != B * 9 or tv.numel() != B * 3 or tuple(ln.shape) != (Bt, L, 6):
            raise ValueError("R (poses * B, 3, 3), t (poses * B, 3), line (B, L, 6) expected")
        g = self.ones if grad_loss is None else _prep(grad_loss, "grad_loss", None, dev)
        lib, s = self._lib, _stream(dev)
        tail_s, fixed_f = self._tail_s, self._fixed_f
        if target_from is not None:
            if target_from is self.st:
                raise ValueError("target_from must be another evaluation's state")
            tws = _target_ws(target_from, B, N, M, L)
            tail_s, fixed_f = tail_s[:-1] + (tws,), fixed_f[:-1] + (tws,)
        self.st.target_state = getattr(target_from, "target_state", None) or target_from  # whose workspace holds cloud 2
        if self.ride is not None:
            self.ride.arm()
        with _guard(dev):
            if RegistrationStep.ONE_CALL:  # forward + backward as one C entry: the backward may ride in the reduce's launch
                check(lib.rrl_registration_step_ex(_p(self.src), _p(Rm), _p(tv), _p(self.tar), _p(ln), *self._head_s, _p(g),
                                                   *tail_s, op, s), "rrl_registration_step")
            else:
                check(lib.rrl_registration_forward_ex(_p(self.src), _p(Rm), _p(tv), _p(self.tar), _p(ln), *fixed_f, op, s),
                      "rrl_registration_forward")
                check(lib.rrl_registration_backward_ex(_p(self.src), _p(Rm), _p(self.tar), *self._fixed_b, _p(g), *self._tail_b, op, s),
                      "rrl_registration_backward")
        if self.prepared:
            self._kept_key = key  # (None for a carried-over target -- it is not built here -- and with keep_target off)
            self._chain_ready = bool(self._chain_left.value & 1)
            self.fused = bool(self._chain_left.value & 2)  # THIS call's records + scans ran as one launch
        self.st.counts_cleared = self._chain_ready
        _IntersectionLoss.last_state = self.st
        if self.ride is not None:  # .chamfer_value: this step's monitor (it rode in the scan's launch, or one launch now)
            _keep_ride(self.st, self.ride)
            self.chamfer_value = chamfer_from_state(self.st)
        return self._out


class LossStep:
    """SURVEY 8(d)'s definition by direct issue (round 4): rigid apply of the source + loss + backward to points1.grad
    (B, N, 9) as ONE C call (rrl_loss_step_ex) on buffers allocated once, no autograd node, no graph -- the drop-in
    chain `tri = rigid_apply(src, R, t); loss = intersection_loss(tri, tar, line); loss.backward()` without its host cost:

        step = ops.LossStep(src_tri, tar_tri, L)                   # once per shape (orders computed here)
        loss, g_points1, info = step(R, t, lines)                  # every iteration; g_points1 = dL/d(moved triangles)
        moved_tri.backward(g_points1)                              # hand it to whatever produced the pose

    R = t = None evaluates the triangles as given (no transform).  Same numbers as the autograd chain: loss bit for bit,
    gradient to the rounding of the scatter's float atomics.  prepared / src_order / tar_order as RegistrationStep."""

    def __init__(self, src_tri, tar_tri, n_lines, rng=(1, 1, 5, 5), transpose_r=True, mode="cull", chunk=0,
                 prepared=None, src_order=None, tar_order=None, chamfer=False, want_payload=False, poses=1, chain=True,
                 deterministic=None):
        """deterministic=True (round 6; include/rrl.h rrl_set_deterministic): points1.grad accumulates in 64-bit fixed point and
        reproduces BIT FOR BIT from call to call (the float atomics of the default scatter agree only to their rounding order);
        the step then runs forward + scatter + a conversion launch (no chain, no riding backward).  None: the library default.
        chain=True (default; round 6, include/rrl.h RRL_F_CHAIN / RRL_F_CHAINED): from its second call on, while the target is
        kept, the step runs the source's records, the target's scan and the source's scan as ONE launch (three launches per
        step instead of four, same loss bits).  A chained step leaves .st.count1 / .st.count2 CLEARED (the per-line stage
        zeroes them behind its read; .st.kj / hs1 / hs2 hold what it read) and does not update .st.status -- info[:, 3] is then
        each sample's OWN NaN flag instead of the batch-wide STATUS[0] --, and its state cannot serve as target_from=.
        chamfer=True: every step also leaves the Chamfer monitor of its clouds in .chamfer_value -- its walk rides in the
        step's scan launch (ChamferRide), as in RegistrationStep.
        want_payload=True: .payload (14,) = [sum of the valid losses, #valid, 0 x 12] after every step -- what a rank
        contributes to the all-reduce of the scalar loss (rrl_hip.dist; points1.grad stays local).
        poses = k > 1: a multi-pose step as RegistrationStep's -- R (k * B, 3, 3), t (k * B, 3) (required), loss (k * B,),
        grad (k * B, N, 9) = dL/d(moved triangles) of every instance."""
        dev = _home(src_tri, tar_tri)
        self.dev = dev
        self.src = _prep(src_tri, "src_tri", 9, dev)
        self.tar = _prep(tar_tri, "tar_tri", 9, dev)
        if self.src.dim() != 3 or self.tar.dim() != 3 or self.src.shape[0] != self.tar.shape[0]:
            raise ValueError("src_tri/tar_tri must be (B, n, 9) with the same B")
        Bt, N, _ = self.src.shape
        M, L = self.tar.shape[1], int(n_lines)
        if Bt == 0 or L <= 0:
            raise ValueError("LossStep needs a non-empty batch and line set")
        self.poses, self.Bt = int(poses), Bt
        if self.poses < 1 or (self.poses > 1 and (mode != "cull" or max(N, M) > 65536)):
            raise ValueError("poses must be >= 1; a multi-pose step runs scan mode cull on clouds <= 65536 triangles")
        B = Bt * self.poses
        prob = Bt if self.poses > 1 else None
        self.ride = ChamferRide(B, N, M, dev) if chamfer else None
        self.chamfer_value = None
        self.dims = (B, N, M, L)
        self.rng = _check_range(rng)
        self.tr, self.mode, self.chunk = int(bool(transpose_r)), _MODES[mode], int(chunk)
        self.st = LossState(B, N, M, L, B, dev)
        self.ones = torch.ones(B, dtype=torch.float32, device=dev)
        self.grad = torch.empty(B, N, 9, dtype=torch.float32, device=dev)
        if prepared is None:
            prepared = os.environ.get("RRL_PREPARED", "1") != "0"
        self.prepared = bool(prepared) and mode == "cull" and max(N, M) <= 65536
        self._kept_key = None
        self.keep_target = True  # see RegistrationStep.invalidate_target
        # (in the workspace's accumulator field: the step's first launch clears it, as for RegistrationStep)
        self.payload = self.st.gacc[B * 12:B * 12 + 14] if want_payload else None
        self._opts = self._opts_kept = make_opts(chamfer=self.ride, payload=self.payload, problems=prob, deterministic=deterministic)  # (None without any)
        # CHAINED steps (round 6; include/rrl.h RRL_F_CHAIN / RRL_F_CHAINED): every prepared step asks the library to leave
        # the workspace's hit counts cleared (the library says through _chain_left whether it did); a step that follows such
        # a step with the target still kept runs records + target scan + source scan as ONE launch.  chain = False turns
        # it off for this object (RRL_CHAIN=0 for the process).
        self.chain = bool(chain)
        self._chain_left = ctypes.c_int32(0)
        self._chain_ready = False  # the previous call on self.st left it chain-clean and nothing has touched it since
        self.fused = False
        if self.prepared:
            self.order1 = _check_order(src_order, Bt, N, dev, "src_order") if src_order is not None else cloud_order(self.src)
            self.order2 = _check_order(tar_order, Bt, M, dev, "tar_order") if tar_order is not None else cloud_order(self.tar)
            kw = dict(order1=self.order1, order2=self.order2, chamfer=self.ride, payload=self.payload, problems=prob,
                      chain_left=self._chain_left, deterministic=deterministic)
            self._opts = make_opts(**kw)  # (.chain = False: no chain flags at all -- the hit counts stay readable)
            self._opts_kept = make_opts(target_kept=True, **kw)
            self._opts_c = (make_opts(chain=_lib.F_CHAIN, **kw), make_opts(target_kept=True, chain=_lib.F_CHAIN, **kw),
                            make_opts(target_kept=True, chain=_lib.F_CHAIN | _lib.F_CHAINED, **kw))
            self._optr_c = tuple(_optr(o) for o in self._opts_c)  # first / kept / chained
        self._optr, self._optr_kept = _optr(self._opts), _optr(self._opts_kept)
        self._lib = _lib.load()
        # what every call hands over / returns unchanged: built once (a LossState field is a fresh view per access: ~6 us of host
        # time per step for `.info` alone)
        self._out = (self.st.loss.view(-1), self.grad, self.st.info)
        self._c_fixed = (_p(self.st.ws), self.st.nbytes, _p(self.st.loss))
        self._c_tail = (_p(self.grad), None, B, N, M, L, self.tr, *self.rng, self.mode, self.chunk, None)
        self._p_ones = _p(self.ones)

    def invalidate_target(self):
        """As RegistrationStep.invalidate_target: the next call rebuilds the target's records."""
        self._kept_key = None

    def __call__(self, R, t, line, grad_loss=None):
        B, N, M, L = self.dims
        dev = self.dev
        Rm = _prep(R, "R", None, dev) if R is not None else None
        tv = _prep(t, "t", None, dev) if t is not None else None
        ln = _prep(line, "line", 6, dev)
        if (Rm is None) != (tv is None) or (Rm is not None and (Rm.numel() != B * 9 or tv.numel() != B * 3)) \
                or tuple(ln.shape) != (self.Bt, L, 6) or (self.poses > 1 and Rm is None):
            raise ValueError("R (poses * B, 3, 3) and t (poses * B, 3) (or both None for poses = 1), line (B, L, 6) expected")
        g = self.ones if grad_loss is None else _prep(grad_loss, "grad_loss", None, dev)
        op, key = self._optr, None
        if self.prepared:
            key = _write_key(self.tar) if self.keep_target else None
            kept = key is not None and key == self._kept_key
            if self.chain:
                op = self._optr_c[2 if (kept and self._chain_ready) else (1 if kept else 0)]
            elif kept:
                op = self._optr_kept
            self._kept_key = None  # (set below, once the call has been issued)
        self._chain_ready = False
        self._chain_left.value = 0
        if self.ride is not None:
            self.ride.arm()
        with _guard(dev):
            check(self._lib.rrl_loss_step_ex(_p(self.src), _p(Rm), _p(tv), _p(self.tar), _p(ln), *self._c_fixed,
                                             self._p_ones if g is self.ones else _p(g), *self._c_tail, op, _stream(dev)),
                  "rrl_loss_step")
        self._kept_key = key
        self._chain_ready = self.prepared and bool(self._chain_left.value & 1)
        self.fused = bool(self._chain_left.value & 2)  # THIS call's records + scans ran as one launch
        self.st.counts_cleared = self._chain_ready
        _IntersectionLoss.last_state = self.st
        if self.ride is not None:
            _keep_ride(self.st, self.ride)
            self.chamfer_value = chamfer_from_state(self.st)
        return self._out


def registration_step_raw(src_tri, R, t, tar_tri, line, rng=(1, 1, 5, 5), transpose_r=True, order1=None, order2=None,
                          chamfer=False):
    """ONE evaluation of the fused training op, forward AND backward for dL/dloss = 1, as one C call on a FRESH state
    (rrl_registration_step_ex: where the tail kernel serves the shape the backward rides in the reduce's launch): returns
    (loss (B,), gR (B, 3, 3), gt (B, 3), info (B, 4), state) -- gR[s], gt[s] = d loss[s] / d(R[s], t[s]); the loss is
    linear in the upstream gradient, so an autograd node scales them (callsites._PackedPoses) instead of running a
    backward of its own.  R, t with k * B_t poses for B_t problems: a multi-pose evaluation (rrl_opts.problems).  Inputs
    must be prepared GPU tensors (fp32, contiguous); scan mode cull.  The outputs live in the returned state's workspace."""
    dev = src_tri.device
    Bt, N, _ = src_tri.shape
    M, L = tar_tri.shape[1], line.shape[1]
    B = R.shape[0]
    if not (tar_tri.shape[0] == line.shape[0] == Bt and t.shape[0] == B) or Bt == 0 or B % Bt or L <= 0:
        raise ValueError("src_tri / tar_tri / line share B_t > 0; R, t hold k * B_t poses; L > 0")
    problems = Bt if B != Bt else 0
    s_m, s_n, e_m, e_n = _check_range(rng)
    st = LossState(B, N, M, L, B, dev)
    ride = ChamferRide(B, N, M, dev) if chamfer else None
    opts = make_opts(order1=_check_order(order1, Bt, N, dev, "order1"), order2=_check_order(order2, Bt, M, dev, "order2"),
                     chamfer=ride, problems=problems)
    gacc = st.gacc
    ones = _ones_cache.get((B, dev))
    if ones is None:
        if len(_ones_cache) > 32:
            _ones_cache.clear()
        ones = _ones_cache[(B, dev)] = torch.ones(B, dtype=torch.float32, device=dev)
    _arm_ride(opts)
    with _guard(dev):
        check(_lib.load().rrl_registration_step_ex(
            _p(src_tri), _p(R), _p(t), _p(tar_tri), _p(line), _p(st.ws), st.nbytes, _p(st.loss), _p(ones),
            _p(gacc[:B * 9]), _p(gacc[B * 9:B * 12]), None, B, N, M, L, int(transpose_r), s_m, s_n, e_m, e_n, SCAN_CULL, 0, None,
            _optr(opts), _stream(dev)), "rrl_registration_step")
    st.target_state = None
    _keep_ride(st, ride)
    _IntersectionLoss.last_state = st
    return st.loss.view(-1), gacc[:B * 9].view(B, 3, 3), gacc[B * 9:B * 12].view(B, 3), st.info, st


_ones_cache = {}


def set_deterministic(on):
    """Bit-reproducible direct backward of registration_loss (fixed-order partial sums, one more tiny
    launch) instead of float atomics; include/rrl.h rrl_set_deterministic.  Process-wide."""
    check(_lib.load().rrl_set_deterministic(int(bool(on))), "rrl_set_deterministic")


def set_reduce_mode(mode):
    """Which reduce kernel the forwards launch: "auto" (tiled where legal and worthwhile), "single", "tiled"
    (include/rrl.h rrl_set_reduce_mode).  Process-wide; same bits either way."""
    check(_lib.load().rrl_set_reduce_mode({"auto": 0, "single": 1, "tiled": 2, "xchg": 3}[mode]), "rrl_set_reduce_mode")


def set_spin_limit(polls):
    """Test hook (include/rrl.h rrl_set_spin_limit): polls before a waiting workgroup of the exchange reduce gives up and
    leaves the sample to its last workgroup's repair; None = the default (2^18)."""
    check(_lib.load().rrl_set_spin_limit(int(1 << 18 if polls is None else polls)), "rrl_set_spin_limit")


def debug_occupy(workgroups, lanes, seconds, stream=None):
    """Test hook (rrl_debug_occupy): a filler launch that holds `workgroups` x `lanes` threads' slots for `seconds` on
    `stream` (a torch.cuda.Stream; default: the current one)."""
    dev = require_gpu()
    raw = ctypes.c_void_p(stream.cuda_stream) if stream is not None else _stream(dev)
    check(_lib.load().rrl_debug_occupy(int(workgroups), int(lanes), int(seconds * 1e8), raw), "rrl_debug_occupy")


def set_sort_parts(parts):
    """Workgroups per cloud of the sort + sphere-tree kernel (0 = automatic; include/rrl.h rrl_set_sort_parts)."""
    check(_lib.load().rrl_set_sort_parts(int(parts)), "rrl_set_sort_parts")


def last_state():
    """LossState of the most recent loss evaluation on this process (workspace views, payload)."""
    return _IntersectionLoss.last_state


def scan_timing(every):
    """Profiling hook: bracket every `every`-th scan launch with HIP events; 0 = off."""
    check(_lib.load().rrl_scan_timing_enable(int(every)), "rrl_scan_timing_enable")


def scan_timing_collect(max_n=1024):
    buf = (ctypes.c_float * max_n)()
    n = _lib.load().rrl_scan_timing_collect(buf, max_n)
    return [float(buf[i]) for i in range(n)]


_counter_buf = None


def scan_counters(on, rows=1 << 17, raw=False):
    """Profiling hook: on=True -> subsequent culled scans run the instrumented kernel, every wavefront writing
    one row of 16 counters into a fresh device buffer of `rows` rows; on=False -> back to the plain kernel.
    Returns the counters of the period that just ended, summed over the rows (int64 [16], see include/rrl.h
    rrl_scan_counters; raw=True: the (rows, 16) table, e.g. for the start / end clocks), or None."""
    global _counter_buf
    prev = _counter_buf
    if on:
        _counter_buf = torch.zeros(int(rows), 16, dtype=torch.int64, device=require_gpu())
        check(_lib.load().rrl_scan_counters(_p(_counter_buf), int(rows)), "rrl_scan_counters")
    else:
        _counter_buf = None
        check(_lib.load().rrl_scan_counters(None, 0), "rrl_scan_counters")
    if prev is None:
        return None
    return prev if raw else prev.sum(0)


_cham_counter_buf = None


def chamfer_counters(on, raw=False, rows=1 << 19):
    """Like scan_counters, for the tree Chamfer (include/rrl.h rrl_chamfer_counters): one 16-slot row per
    wavefront of the walk (plain stores; rows beyond `rows` are dropped by the kernel), summed here (raw=True:
    the table of rows).  A full table needs 8 * 2 B * ceil(max(N, M) / 64) rows."""
    global _cham_counter_buf
    prev = _cham_counter_buf
    _cham_counter_buf = torch.zeros(16 * int(rows), dtype=torch.int64, device=require_gpu()) if on else None
    check(_lib.load().rrl_chamfer_counters(_p(_cham_counter_buf), int(rows) if on else 0), "rrl_chamfer_counters")
    if prev is None:
        return None
    return prev.reshape(-1, 16) if raw else prev.reshape(-1, 16).sum(0)


# ---------------------------------------------------------------------------------------
class _RigidApply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, R, t, transpose_r, channel_first):
        dev = _home(x, R, t)
        xs = _prep(x, "x", None, dev)
        Rm, tv = _prep(R, "R", None, dev), _prep(t, "t", None, dev)
        B = Rm.shape[0]
        n = xs.numel() // (3 * B)
        y = torch.empty_like(xs)
        _run(dev, "rrl_rigid_apply_fwd", _p(xs), _p(Rm), _p(tv), _p(y), B, n, int(transpose_r), int(channel_first))
        ctx.save_for_backward(xs, Rm)
        ctx.meta = (B, n, int(transpose_r), int(channel_first), x.device, R.device, t.device,
                    R.shape, t.shape)
        return y.to(x.device)

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        xs, Rm = ctx.saved_tensors
        B, n, tr, cf, xdev, Rdev, tdev, Rshape, tshape = ctx.meta
        g = gy.detach().to(device=xs.device, dtype=torch.float32).contiguous()
        gx = torch.empty_like(xs) if ctx.needs_input_grad[0] else None
        gR = torch.empty(B, 3, 3, device=xs.device)
        gt = torch.empty(B, 3, device=xs.device)
        nblk = lib.rrl_rigid_bwd_blocks(n)
        partial = torch.empty(B, max(nblk, 1), 12, device=xs.device)
        _run(xs.device, "rrl_rigid_apply_bwd", _p(xs), _p(Rm), _p(g), _p(gx), _p(gR), _p(gt), _p(partial), B, n, tr, cf)
        return (gx.to(xdev) if gx is not None else None, gR.reshape(Rshape).to(Rdev),
                gt.reshape(tshape).to(tdev), None, None)


def rigid_apply(x, R, t, transpose_r=False, channel_first=False):
    """y = x R + t (transpose_r=False) or y = x R^T + t; x is (B, n, 3) or (B, 3, n)
    (channel_first); R (B,3,3), t (B,3).  Differentiable in x, R and t."""
    R3 = R.reshape(-1, 3, 3)
    B = R3.shape[0]
    if x.numel() % (3 * B) != 0:
        raise ValueError("x does not divide into B point sets")
    return _RigidApply.apply(x, R3, t.reshape(B, 3), transpose_r, channel_first)


# ---------------------------------------------------------------------------------------
class _Se3Exp(torch.autograd.Function):
    """se(3) exponential map xi (B,6) -> R (B,3,3), T (B,3) as ONE launch each way
    (code/LieAlgebra/se3.py:83-106; as torch ops it is ~40 tiny kernels forward, ~100 backward)."""

    @staticmethod
    def forward(ctx, xi):
        x = _prep(xi, "xi").reshape(-1, 6)
        B = x.shape[0]
        R = torch.empty(B, 3, 3, device=x.device)
        T = torch.empty(B, 3, device=x.device)
        _run(x.device, "rrl_se3_exp", _p(x), _p(R), _p(T), B)
        ctx.save_for_backward(x)
        ctx.meta = (xi.shape, xi.device)
        return R, T

    @staticmethod
    def backward(ctx, gR, gT):
        (x,) = ctx.saved_tensors
        shape, dev = ctx.meta
        B = x.shape[0]
        gRc = _prep(gR, "gR", None, x.device) if gR is not None else None
        gTc = _prep(gT, "gT", None, x.device) if gT is not None else None
        gxi = torch.empty(B, 6, device=x.device)
        _run(x.device, "rrl_se3_exp_bwd", _p(x), _p(gRc), _p(gTc), _p(gxi), B)
        return gxi.reshape(shape).to(dev)


def se3_exp(xi):
    """(R (B,3,3), T (B,3)) = exp3(xi) on the GPU, differentiable in xi; xi is (6,) or (B,6)."""
    return _Se3Exp.apply(xi)


def adam_gated(param, grad, m, v, state, lr, gate=None, betas=(0.9, 0.999), eps=1e-8):
    """In-place torch.optim.Adam step on `param` (fp32, contiguous, GPU) with device-side scalars:
    state (1,) = step count, lr (1,) ; skipped when gate (int32 tensor) has gate[0] <= 0."""
    n = param.numel()
    _run(param.device, "rrl_adam_gated", _p(param), _p(grad), _p(m), _p(v), _p(state), _p(lr), _p(gate), n,
         float(betas[0]), float(betas[1]), float(eps))


# ---------------------------------------------------------------------------------------
CHAMFER_TREE = True   # False: the brute-force kernel (every pair evaluated); same keys, same value
_chamfer_ws_bytes = {}


class _Chamfer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, order_x=None, order_y=None):
        dev = _home(x, y)
        xs, ys = _prep(x, "points_x", 3, dev), _prep(y, "points_y", 3, dev)
        if xs.dim() != 3 or ys.dim() != 3 or xs.shape[0] != ys.shape[0]:
            raise ValueError("chamfer_dist expects (B, M, 3) and (B, N, 3)")
        B, N, _ = xs.shape
        M = ys.shape[1]
        bx = torch.empty(B, N, dtype=torch.int64, device=xs.device)
        by = torch.empty(B, M, dtype=torch.int64, device=xs.device)
        val = torch.empty(1, device=xs.device)
        if CHAMFER_TREE and 0 < B <= 32767 and 0 < max(N, M) <= 65536 and min(N, M) > 0:
            # sorted clouds + sphere tree + pruned walk (rrl_chamfer.hip): keys identical to brute force
            nb = _chamfer_ws_bytes.get((B, N, M))
            if nb is None:
                nb = _chamfer_ws_bytes[(B, N, M)] = int(_lib.load().rrl_chamfer_workspace_bytes(B, N, M))
            ws = torch.empty(nb, dtype=torch.uint8, device=dev)
            if order_x is not None and order_y is not None:  # prepared clouds: no sort in this call
                ox, oy = _check_order(order_x, B, N, dev, "order_x"), _check_order(order_y, B, M, dev, "order_y")
                _run(dev, "rrl_chamfer_tree_fwd_ex", _p(xs), _p(ys), _p(ws), nb, _p(bx), _p(by), _p(val), B, N, M, _p(ox), _p(oy),
                     None, 0)
            else:
                _run(dev, "rrl_chamfer_tree_fwd", _p(xs), _p(ys), _p(ws), nb, _p(bx), _p(by), _p(val), B, N, M)
        else:
            _run(dev, "rrl_chamfer_fwd", _p(xs), _p(ys), _p(bx), _p(by), _p(val), B, N, M)
        ctx.save_for_backward(xs, ys, bx, by)
        ctx.devs = (x.device, y.device)
        return val.reshape(()).to(x.device)

    @staticmethod
    def backward(ctx, gval):
        xs, ys, bx, by = ctx.saved_tensors
        B, N, _ = xs.shape
        M = ys.shape[1]
        g = gval.detach().to(device=xs.device, dtype=torch.float32).reshape(1).contiguous()
        gx = torch.zeros_like(xs) if ctx.needs_input_grad[0] else None
        gy = torch.zeros_like(ys) if ctx.needs_input_grad[1] else None
        _run(xs.device, "rrl_chamfer_bwd", _p(xs), _p(ys), _p(bx), _p(by), _p(g), _p(gx), _p(gy), B, N, M)
        return (gx.to(ctx.devs[0]) if gx is not None else None,
                gy.to(ctx.devs[1]) if gy is not None else None, None, None)


def chamfer(x, y, order_x=None, order_y=None):
    """chamfer_dist (code/loss.py:236-252) of x (B, N, 3), y (B, M, 3): scalar, differentiable.  order_x / order_y:
    ops.cloud_order of the two point clouds (or of the pseudo-triangles they are the first points of) in any rigid pose
    -- both given, the per-call sort is skipped (same keys, same value)."""
    return _Chamfer.apply(x, y, order_x, order_y)


def chamfer_from_state(state=None, keys=False):
    """Chamfer distance between the two clouds of a loss evaluation (default: the latest one) from the
    sorted records and sphere trees that evaluation left in its workspace -- no second sort
    (include/rrl.h rrl_chamfer_from_loss; two launches instead of three).  The "points" are the first
    points of the pseudo-triangles (code/loss.py:473-485: row = [P, neighbour, neighbour]); for the fused
    op cloud 1 is the MOVED source.  Equals chamfer(P0 of cloud 1, P0 of cloud 2): use it for the trainers'
    monitor next to the loss when their point sets are those first points.  Not differentiable.
    keys=True also returns the (B, N) / (B, M) u64 keys (distance bits << 32 | argmin)."""
    st = state or _IntersectionLoss.last_state
    if st is None:
        raise ValueError("no loss evaluation to take the clouds from")
    ride = getattr(st, "cham_ride", None)
    if ride is not None:  # the walk rode in this evaluation's own scan launch (ChamferRide): nothing left to launch
        return (ride.val.reshape(()), ride.bx, ride.by) if keys else ride.val.reshape(())
    B, N, M, L, _ = st.dims
    dev = st.ws.device
    tar = getattr(st, "target_state", None) or st
    if tuple(tar.dims[:4]) != (B, N, M, L) or tar.ws.device != dev:
        raise ValueError("the carried-over target state does not match")
    nb = _chamfer_ws_bytes.get((B, N, M))
    if nb is None:
        nb = _chamfer_ws_bytes[(B, N, M)] = int(_lib.load().rrl_chamfer_workspace_bytes(B, N, M))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    bx = torch.empty(B, N, dtype=torch.int64, device=dev)
    by = torch.empty(B, M, dtype=torch.int64, device=dev)
    val = torch.empty(1, device=dev)
    _run(dev, "rrl_chamfer_from_loss", _p(st.ws), _p(tar.ws), st.nbytes, B, N, M, L, _p(ws), nb, _p(bx), _p(by),
         _p(val))
    return (val.reshape(()), bx, by) if keys else val.reshape(())


def chamfer_group_means(state=None, groups=1):
    """The Chamfer monitor of a MULTI-POSE evaluation per pose: (groups,) means over the evaluation's instances
    [g * B_t, (g + 1) * B_t) -- what chamfer_from_state returns for each pose evaluated on its own.  From the
    per-(sample, direction) sums the walk left in its workspace (include/rrl.h rrl_chamfer_group_means: one tiny launch) --
    the walk that rode in the evaluation's scan launch, else one run now."""
    st = state or _IntersectionLoss.last_state
    if st is None:
        raise ValueError("no loss evaluation to take the clouds from")
    B, N, M, L, _ = st.dims
    if B % groups:
        raise ValueError("groups must divide the evaluation's instances")
    dev = st.ws.device
    ride = getattr(st, "cham_ride", None)
    if ride is not None:
        ws = ride.ws
    else:
        tar = getattr(st, "target_state", None) or st
        nb = _chamfer_ws_bytes.get((B, N, M))
        if nb is None:
            nb = _chamfer_ws_bytes[(B, N, M)] = int(_lib.load().rrl_chamfer_workspace_bytes(B, N, M))
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        bx = torch.empty(B, N, dtype=torch.int64, device=dev)
        by = torch.empty(B, M, dtype=torch.int64, device=dev)
        val = torch.empty(1, device=dev)
        _run(dev, "rrl_chamfer_from_loss", _p(st.ws), _p(tar.ws), st.nbytes, B, N, M, L, _p(ws), nb, _p(bx), _p(by), _p(val))
    out = torch.empty(groups, device=dev)
    _run(dev, "rrl_chamfer_group_means", _p(ws), ws.numel(), _p(out), int(groups), B, N, M)
    return out


# ---------------------------------------------------------------------------------------
def dense_scan(tri, line):
    """(norm_d (B*L, N, 3) float32, label (B, L, N) bool, status (1,) int32) -- the dense tables of
    code/loss.py:68-112 (rrl_dense_scan); B*L*N*13 bytes of output, so mind the sizes."""
    dev = _home(tri, line)
    t, ln = _prep(tri, "point_neis", 9, dev), _prep(line, "line", 6, dev)
    B, N, _ = t.shape
    L = ln.shape[1]
    norm_d = torch.empty(B * L, N, 3, dtype=torch.float32, device=t.device)
    label = torch.empty(B, L, N, dtype=torch.uint8, device=t.device)
    status = torch.empty(1, dtype=torch.int32, device=t.device)
    _run(dev, "rrl_dense_scan", _p(t), _p(ln), _p(norm_d), _p(label), _p(status), B, N, L)
    return norm_d, label.view(torch.bool), status


def aabb(v):
    """(B, n, 3) -> (B, 6) = min xyz, max xyz on the GPU."""
    vs = _prep(v, "vertices", 3)
    B, n, _ = vs.shape
    out = torch.empty(B, 6, device=vs.device)
    _run(vs.device, "rrl_aabb", _p(vs), _p(out), B, n)
    return out


def box_accept(lines, aabb1, aabb2):
    """The resampler's accept test on given lines (B, n, 6) against two AABBs (B, 6) each (ops.aabb):
    returns (mask (B, n) uint8 -- bit 0 / 1: box 1 / 2 crossed by the reference's sub-area test, bit 2:
    the conservative slab pre-test passes; accepted == (mask & 3) == 3 --, hits (B, n, 2) int32 =
    the reference's label1, label2 counts)."""
    dev = _home(lines, aabb1, aabb2)
    ln = _prep(lines, "lines", 6, dev)
    b1, b2 = _prep(aabb1, "aabb1", 6, dev), _prep(aabb2, "aabb2", 6, dev)
    B, n, _ = ln.shape
    mask = torch.empty(B, n, dtype=torch.uint8, device=dev)
    hits = torch.empty(B, n, 2, dtype=torch.int32, device=dev)
    _run(dev, "rrl_box_accept", _p(ln), _p(b1), _p(b2), _p(mask), _p(hits), B, n)
    return mask, hits


_sampler_rng = {}  # device index -> int64[4] state of the library's generator (include/rrl.h rrl_sample_lines_rng)
_sampler_key = {}  # device index -> (seed, offset) of torch's CUDA generator at the last look (never written back)
_sampler_pin = {}  # device index -> pinned int64[4] staging buffer of a (re)seed


def sampler_rng(dev=None, seed=None):
    """State of the library's own uniform generator for the line sampler on `dev` (Philox4x32-10 inside the sampler
    kernels: [seed, call counter, ticket, 0]).  The counter lives on the device and is advanced by every call, so a
    captured step replays a fresh stream each time with no host-side bookkeeping.
    Seeding follows torch WITHOUT touching it (round 4): the state is (re)created from the SEED of torch's CUDA generator
    of `dev` whenever torch was re-seeded since the last look -- its seed changed, or its offset went BACKWARDS (only
    torch.manual_seed / torch.cuda.manual_seed / set_state do that) -- torch's offset is read, never moved, so other
    GPU draws (dropout, rand) neither re-seed the sampler nor are shifted by it.  `torch.manual_seed(s)` followed by the
    same calls gives the same lines.  One blind spot: re-seeding torch with the SAME seed while nothing else has drawn
    from its CUDA generator is invisible (seed and offset unchanged) -- use seed=<int> to re-seed explicitly then.
    The new state is uploaded from pinned memory without blocking the host.
    (No check while the stream is capturing: a captured step keeps the state it was captured with.)"""
    dev = dev if dev is not None else require_gpu()
    st = _sampler_rng.get(dev.index)
    if seed is None and not (st is not None and torch.cuda.is_current_stream_capturing()):
        g = torch.cuda.default_generators[dev.index]
        tseed, toff = g.initial_seed(), g.get_offset()
        known = _sampler_key.get(dev.index)
        if st is None or known is None or tseed != known[0] or toff < known[1]:
            seed = (tseed ^ 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        _sampler_key[dev.index] = (tseed, toff)
    elif seed is not None and not torch.cuda.is_current_stream_capturing():
        g = torch.cuda.default_generators[dev.index]  # an explicit seed stands until torch is re-seeded AFTER it
        _sampler_key[dev.index] = (g.initial_seed(), g.get_offset())
    if st is None or seed is not None:
        sd = int(seed) & 0x7FFFFFFFFFFFFFFF
        host = _sampler_pin.get(dev.index)
        if host is None:
            host = _sampler_pin[dev.index] = torch.zeros(4, dtype=torch.int64).pin_memory()
        elif st is not None:
            torch.cuda.current_stream(dev).synchronize()  # (re-seeding only: the previous upload must have left the pinned buffer)
        host[0], host[1], host[2], host[3] = sd, 0, 0, 0
        if st is None:
            st = _sampler_rng[dev.index] = torch.empty(4, dtype=torch.int64, device=dev)
        st.copy_(host, non_blocking=True)  # in place: a captured step keeps pointing at the same memory
    return st


def sample_lines(rands, r, centers, aabb1, aabb2, out=None, rng_shape=None):
    """rands (rounds, 4, B, n) uniform draws -> lines (B, n, 6), filled (B,) int32.
    rands=None with rng_shape=(rounds, B, n): the uniforms are drawn inside the kernels by the library's generator
    (sampler_rng) -- no (rounds, 4, B, n) tensor exists at all.
    out: a contiguous fp32 (B, n, 6) GPU tensor to write the lines into (no extra copy)."""
    dev = _home(out, rands, aabb1, aabb2, r)
    if rands is not None:
        rd = _prep(rands, "rands", None, dev)
        rounds, four, B, n = rd.shape
        assert four == 4
    else:
        rounds, B, n = (int(v) for v in rng_shape)
    rr = _prep(r, "r", None, dev).reshape(B)
    cc = _prep(centers, "centers", None, dev).reshape(B, 3)
    aabb1 = _prep(aabb1, "aabb1", 6, dev) if aabb1 is not None else None
    aabb2 = _prep(aabb2, "aabb2", 6, dev) if aabb2 is not None else None
    if out is None:
        lines = torch.empty(B, n, 6, device=dev)
    else:
        if not (out.is_cuda and out.device == dev and out.dtype == torch.float32 and out.is_contiguous()
                and out.numel() == B * n * 6):
            raise ValueError("out must be a contiguous fp32 GPU tensor of B * n * 6 elements")
        lines = out
    filled = torch.empty(B, dtype=torch.int32, device=dev)
    scratch = torch.empty(B * max(rounds, 1) * ((n + 1023) // 1024) * 32, dtype=torch.int32, device=dev)
    if rands is not None:
        _run(dev, "rrl_sample_lines", _p(rd), _p(rr), _p(cc), _p(aabb1), _p(aabb2), _p(lines), _p(filled), _p(scratch),
             B, n, rounds)
    else:
        _run(dev, "rrl_sample_lines_rng", _p(sampler_rng(dev)), _p(rr), _p(cc), _p(aabb1), _p(aabb2), _p(lines), _p(filled),
             _p(scratch), B, n, rounds)
    if out is not None:
        _touched(out)
    return lines, filled


def rigid_apply_into(x, R, t, out, transpose_r=False):
    """out[...] = x R + t (or x R^T + t) without autograd and without a temporary: x, out contiguous
    fp32 (B, n, 3) on the GPU."""
    Rm, tv = _prep(R, "R", None, x.device).reshape(-1, 3, 3), _prep(t, "t", None, x.device).reshape(-1, 3)
    B = Rm.shape[0]
    n = x.numel() // (3 * B)
    _run(x.device, "rrl_rigid_apply_fwd", _p(x), _p(Rm), _p(tv), _p(out), B, n, int(transpose_r), 0)
    _touched(out)
    return out


def rigid_apply_aabb_into(x, R, t, out, box, transpose_r=False):
    """out[...] = x R + t and box (B, 6) = the AABB of out, one launch (rrl_rigid_apply_aabb): x, out contiguous fp32
    (B, n, 3) on the GPU."""
    Rm, tv = _prep(R, "R", None, x.device).reshape(-1, 3, 3), _prep(t, "t", None, x.device).reshape(-1, 3)
    B = Rm.shape[0]
    n = x.numel() // (3 * B)
    _run(x.device, "rrl_rigid_apply_aabb", _p(x), _p(Rm), _p(tv), _p(out), _p(box), B, n, int(transpose_r))
    _touched(out, box)
    return out


def se3_adam_step(xi, gR, gT, m, v, state, lr, gate, R, T, *, gxi=None, loss=None, value=None, table=None,
                  cursor=None, row=None, betas=(0.9, 0.999), eps=1e-8, aabb_rows=None, box=None):
    """One pose's backward through the exponential, gated Adam step, exponential of the updated xi into (R, T) and
    the log row, in one launch (rrl_se3_adam_step): all tensors contiguous fp32 on the GPU (cursor int64, gate int32),
    xi (6,), R (.., 3, 3), T (.., 3).  aabb_rows (rows, 8) + box (6,): also the AABB over the rows' (min xyz, max xyz)
    -- the loss state's `apart[0, 0, :ceil(N / 256)]` gives the moved source's box for the next epoch's sampler."""
    nrows = 0 if table is None else table.shape[0]
    _run(xi.device, "rrl_se3_adam_step", _p(xi), _p(gR), _p(gT), _p(m), _p(v), _p(state), _p(lr), _p(gate),
         float(betas[0]), float(betas[1]), float(eps), _p(R), _p(T), _p(gxi), _p(loss), _p(value), _p(table),
         _p(cursor), nrows, _p(row), _p(aabb_rows), 0 if aabb_rows is None else int(aabb_rows.shape[0]), _p(box))


def log_row(loss, value, info, table, cursor, row=None):
    """table[cursor[0]] = (loss[0], value[0], info[0] > 0); cursor[0] += 1 -- one launch, on the device."""
    _run(table.device, "rrl_log_row", _p(loss), _p(value), _p(info), _p(table), _p(cursor), table.shape[0], _p(row))
