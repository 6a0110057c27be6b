"""ctypes front-end of the CPU oracle (oracle/rrl_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of rrl_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package (a-robust-registration-loss_amd/) never does.

Parity pin: tests/test_oracle_golden.py checks every function here against
vectors captured from the reference implementation by
tests/golden/make_golden.py (run in the build container, where
/root/reference exists).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librrl_oracle.so")


def build(force=False):
    """Compile librrl_oracle.so with the committed Makefile (gcc, -ffp-contract=off)."""
    src = os.path.join(_HERE, "rrl_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "librrl_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


class OracleInfo(ctypes.Structure):
    _fields_ = [("n_selected", ctypes.c_int32), ("n_buckets", ctypes.c_int32),
                ("n_values", ctypes.c_int32), ("nan_flag", ctypes.c_int32),
                ("median", ctypes.c_float), ("loss", ctypes.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=ctypes.c_float):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


def tri_threshold(tri):
    """tri (N,9) -> thr (N,)   [code/loss.py:94-110]"""
    tri = _f(tri).reshape(-1, 9)
    thr = np.empty(tri.shape[0], np.float32)
    lib().rrl_oracle_tri_threshold(_p(tri), ctypes.c_int(tri.shape[0]), _p(thr))
    return thr


def scan(tri, line, cap=8, want_label=False):
    """Dense scan of one cloud.  Returns dict(count, hit_idx, hit_w, label, nan)."""
    tri = _f(tri).reshape(-1, 9)
    line = _f(line).reshape(-1, 6)
    N, L = tri.shape[0], line.shape[0]
    count = np.zeros(L, np.int32)
    hit_idx = np.full((L, cap), -1, np.int32)
    hit_w = np.zeros((L, cap, 3), np.float32)
    label = np.zeros((L, N), np.uint8) if want_label else None
    nan = ctypes.c_int(0)
    lib().rrl_oracle_scan(_p(tri), ctypes.c_int(N), _p(line), ctypes.c_int(L),
                          _p(count, ctypes.c_int32), _p(hit_idx, ctypes.c_int32), _p(hit_w),
                          ctypes.c_int(cap), _p(label, ctypes.c_uint8) if want_label else None,
                          ctypes.byref(nan))
    return dict(count=count, hit_idx=hit_idx, hit_w=hit_w,
                label=label.astype(bool) if want_label else None, nan=bool(nan.value))


def loss(tri1, tri2, line, rng=(1, 1, 5, 5), grad_out=1.0, want_grad=True, want_grad2=False,
         want_D=False):
    """One-sample loss + closed-form gradient.  Returns dict; loss is None when no
    bucket is populated (reference: (None, None, None), code/loss.py:231-232)."""
    tri1 = _f(tri1).reshape(-1, 9)
    tri2 = _f(tri2).reshape(-1, 9)
    line = _f(line).reshape(-1, 6)
    N, M, L = tri1.shape[0], tri2.shape[0], line.shape[0]
    g1 = np.zeros((N, 9), np.float32) if want_grad else None
    g2 = np.zeros((M, 9), np.float32) if want_grad2 else None
    cap = L * 64 if want_D else 0
    D = np.zeros(max(cap, 1), np.float32) if want_D else None
    info = OracleInfo()
    rc = lib().rrl_oracle_loss(*[ctypes.c_int(int(v)) for v in rng], _p(tri1), ctypes.c_int(N),
                               _p(tri2), ctypes.c_int(M), _p(line), ctypes.c_int(L),
                               ctypes.c_float(grad_out), _p(g1), _p(g2), _p(D),
                               ctypes.c_int(cap), ctypes.byref(info))
    if rc < 0:
        raise ValueError("bucket range outside the oracle's supported 1..8")
    return dict(loss=(np.float32(info.loss) if rc == 0 else None), grad1=g1, grad2=g2,
                D=(D[:info.n_values] if want_D else None), median=np.float32(info.median),
                n_selected=info.n_selected, n_buckets=info.n_buckets, n_values=info.n_values,
                nan=bool(info.nan_flag))


def chamfer_parts(x, y):
    """x (N,3), y (M,3) -> (min_x, arg_x, min_y, arg_y)   [code/loss.py:38-52, 236-252]"""
    x = _f(x).reshape(-1, 3)
    y = _f(y).reshape(-1, 3)
    N, M = x.shape[0], y.shape[0]
    mx, my = np.empty(N, np.float32), np.empty(M, np.float32)
    ax, ay = np.empty(N, np.int32), np.empty(M, np.int32)
    lib().rrl_oracle_chamfer(_p(x), ctypes.c_int(N), _p(y), ctypes.c_int(M), _p(mx),
                             _p(ax, ctypes.c_int32), _p(my), _p(ay, ctypes.c_int32))
    return mx, ax, my, ay


def chamfer(x, y):
    """Batched chamfer scalar: mean over all B*(N+M) minima (code/loss.py:250-251)."""
    x = _f(x)
    y = _f(y)
    vals = []
    for b in range(x.shape[0]):
        mx, _, my, _ = chamfer_parts(x[b], y[b])
        vals.append(mx)
    for b in range(x.shape[0]):
        mx, _, my, _ = chamfer_parts(x[b], y[b])
        vals.append(my)
    return np.float32(np.mean(np.concatenate(vals).astype(np.float64)))


def rigid_apply(x, R, T, transpose_r=False):
    x = _f(x).reshape(-1, 3)
    R = _f(R).reshape(3, 3)
    T = _f(T).reshape(3)
    y = np.empty_like(x)
    lib().rrl_oracle_rigid_apply(_p(x), ctypes.c_int(x.shape[0]), _p(R), _p(T),
                                 ctypes.c_int(int(transpose_r)), _p(y))
    return y


def bbox(v):
    """(n,3) -> (8,3) AABB corners in the reference order (code/loss.py:325-351)."""
    v = _f(v).reshape(-1, 3)
    out = np.empty((8, 3), np.float32)
    lib().rrl_oracle_bbox(_p(v), ctypes.c_int(v.shape[0]), _p(out))
    return out


def box_hits(bb, lines):
    """Number of box triangles each line crosses per the reference's area test."""
    bb = _f(bb).reshape(8, 3)
    lines = _f(lines).reshape(-1, 6)
    fn = lib().rrl_oracle_box_hits
    fn.restype = ctypes.c_int
    return np.array([fn(_p(bb), _p(lines[i])) for i in range(lines.shape[0])], np.int32)


def make_lines(a1, u1, a2, u2, r, center):
    a1, u1, a2, u2 = (_f(v).reshape(-1) for v in (a1, u1, a2, u2))
    center = _f(center).reshape(3)
    out = np.empty((a1.shape[0], 6), np.float32)
    lib().rrl_oracle_make_lines(_p(a1), _p(u1), _p(a2), _p(u2), ctypes.c_int(a1.shape[0]),
                                ctypes.c_float(float(r)), _p(center), _p(out))
    return out


def resample_lines(rands, r, center, v1, v2, n_lines):
    """10-round rejection sampler for ONE sample (code/loss.py:415-432, 365-381).
    rands: array (rounds, 4, n_lines) of the uniform draws in reference order
    (alpha1, u1, alpha2, u2).  Unfilled rows stay zero."""
    bb1, bb2 = bbox(v1), bbox(v2)
    out = np.zeros((n_lines, 6), np.float32)
    filled = 0
    for rd in range(rands.shape[0]):
        cand = make_lines(rands[rd, 0], rands[rd, 1], rands[rd, 2], rands[rd, 3], r, center)
        ok = (box_hits(bb1, cand) * box_hits(bb2, cand)) > 0
        keep = cand[ok]
        if filled > n_lines:  # reference quirk: `counter > N` skips, `== N` still enters
            continue
        take = keep[: max(0, n_lines - filled)]
        out[filled:filled + take.shape[0]] = take
        filled += keep.shape[0]
    return out


# --- se(3) exponential map, numpy fp32 (code/LieAlgebra/se3.py:83-106, sinc.py) ---

def _sincs(t):
    t = np.float32(t)
    if abs(t) < np.float32(0.01):
        t2 = t * t
        s1 = 1 - t2 / 6 * (1 - t2 / 20 * (1 - t2 / 42))
        s2 = 0.5 * (1 - t2 / 12 * (1 - t2 / 30 * (1 - t2 / 56)))
        s3 = 1 / 6 * (1 - t2 / 20 * (1 - t2 / 42 * (1 - t2 / 72)))
    else:
        s1 = np.sin(t) / t
        s2 = (1 - np.cos(t)) / (t * t)
        s3 = (t - np.sin(t)) / (t ** 3)
    return np.float32(s1), np.float32(s2), np.float32(s3)


def exp3(xi):
    """xi (6,) -> R (3,3), p (3,) in float64-accumulated numpy (tolerance oracle)."""
    xi = np.asarray(xi, np.float32).astype(np.float64)
    w, v = xi[:3], xi[3:]
    t = np.float32(np.sqrt(np.sum(w * w)))
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    S = W @ W
    s1, s2, s3 = (float(s) for s in _sincs(t))
    R = np.eye(3) + s1 * W + s2 * S
    V = np.eye(3) + s2 * W + s3 * S
    return R.astype(np.float32), (V @ v).astype(np.float32)
