"""ctypes binding of librrl_hip.so (C ABI: include/rrl.h).

There is NO fallback: if the library is missing, or no MI355X is visible, every op raises.
"""
import ctypes
import os

import torch  # noqa: F401  -- FIRST: librrl_hip.so must bind to the libamdhip64 PyTorch loaded,
#                              or its streams / device pointers belong to another HIP runtime

from .build import LIB

_c = ctypes
_P = _c.c_void_p
_I = _c.c_int

# name -> argtypes (all return int unless noted)
_Z = _c.c_size_t
_SIGS = {
    "rrl_workspace_layout": [_I, _I, _I, _I, _P],
    "rrl_loss_forward": [_P, _P, _P, _P, _Z, _P] + [_I] * 11 + [_P],
    "rrl_loss_backward": [_P, _P, _P, _Z, _P, _P, _P] + [_I] * 5 + [_P],
    "rrl_registration_forward": [_P] * 6 + [_Z, _P] + [_I] * 11 + [_P],
    "rrl_registration_forward_cached": [_P] * 6 + [_Z, _P] + [_I] * 11 + [_P, _P],
    "rrl_loss_forward_cached": [_P, _P, _P, _P, _Z, _P] + [_I] * 11 + [_P, _P],
    "rrl_loss_forward_info": [_P, _P, _P, _P, _Z, _P] + [_I] * 11 + [_P, _P, _P],
    "rrl_registration_backward": [_P] * 4 + [_Z] + [_P] * 6 + [_I] * 5 + [_P],
    "rrl_registration_step": [_P] * 6 + [_Z] + [_P] * 5 + [_I] * 11 + [_P, _P],
    "rrl_loss_forward_ex": [_P, _P, _P, _P, _Z, _P] + [_I] * 11 + [_P, _P, _P],
    "rrl_registration_forward_ex": [_P] * 6 + [_Z, _P] + [_I] * 11 + [_P, _P, _P],
    "rrl_registration_backward_ex": [_P] * 4 + [_Z] + [_P] * 6 + [_I] * 5 + [_P, _P],
    "rrl_registration_step_ex": [_P] * 6 + [_Z] + [_P] * 5 + [_I] * 11 + [_P, _P, _P],
    "rrl_loss_step_ex": [_P] * 6 + [_Z] + [_P] * 4 + [_I] * 11 + [_P, _P, _P],
    "rrl_cloud_order": [_P, _P, _P, _Z, _I, _I, _P],
    "rrl_cloud_order_points": [_P, _P, _P, _Z, _I, _I, _P],
    "rrl_tri_prepare_ex": [_P, _P, _P, _Z, _I, _I, _I, _I, _P, _P],
    "rrl_line_tri_scan_ex": [_P, _P, _Z] + [_I] * 6 + [_P, _P],
    "rrl_loss_reduce_ex": [_P, _Z, _P] + [_I] * 9 + [_P, _P],
    "rrl_line_pair_dist_ex": [_P, _P, _P, _P, _Z] + [_I] * 9 + [_P, _P],
    "rrl_tri_prepare": [_P, _P, _P, _Z, _I, _I, _I, _I, _P],
    "rrl_line_tri_scan": [_P, _P, _Z] + [_I] * 6 + [_P],
    "rrl_line_pair_dist": [_P, _P, _P, _P, _Z] + [_I] * 9 + [_P],
    "rrl_loss_reduce": [_P, _Z, _P] + [_I] * 9 + [_P],
    "rrl_loss_reduce_rows": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "rrl_shard_payload": [_P, _P, _Z, _P, _P, _P, _I, _I, _I, _I, _P],
    "rrl_set_scan_variant": [_I],
    "rrl_demo_epoch": [_P, _P],
    "rrl_set_spin_limit": [_c.c_longlong],
    "rrl_debug_occupy": [_I, _I, _c.c_longlong, _P],
    "rrl_set_deterministic": [_I],
    "rrl_set_reduce_mode": [_I],
    "rrl_set_sort_parts": [_I],
    "rrl_scan_timing_enable": [_I],
    "rrl_scan_timing_collect": [_P, _I],
    "rrl_scan_counters": [_P, _c.c_longlong],
    "rrl_chamfer_counters": [_P, _c.c_longlong],
    "rrl_chamfer_group_means": [_P, _Z, _P, _I, _I, _I, _I, _P],
    "rrl_rigid_apply_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "rrl_rigid_bwd_blocks": [_I],
    "rrl_rigid_apply_bwd": [_P] * 7 + [_I] * 4 + [_P],
    "rrl_chamfer_fwd": [_P] * 5 + [_I] * 3 + [_P],
    "rrl_chamfer_bwd": [_P] * 7 + [_I] * 3 + [_P],
    "rrl_chamfer_tree_fwd": [_P, _P, _P, _Z, _P, _P, _P, _I, _I, _I, _P],
    "rrl_chamfer_from_loss": [_P, _P, _Z, _I, _I, _I, _I, _P, _Z, _P, _P, _P, _P],
    "rrl_chamfer_tree_fwd_ex": [_P, _P, _P, _Z, _P, _P, _P, _I, _I, _I, _P, _P, _P, _c.c_longlong, _P],
    "rrl_chamfer_from_loss_ex": [_P, _P, _Z, _I, _I, _I, _I, _P, _Z, _P, _P, _P, _P, _c.c_longlong, _P],
    "rrl_aabb": [_P, _P, _I, _I, _P],
    "rrl_box_accept": [_P, _P, _P, _P, _P, _I, _I, _P],
    "rrl_log_row": [_P, _P, _P, _P, _P, _c.c_longlong, _P, _P],
    "rrl_se3_adam_step": [_P] * 8 + [_c.c_double] * 3 + [_P] * 7 + [_c.c_longlong, _P, _P, _I, _P, _P],
    "rrl_rigid_apply_aabb": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "rrl_se3_exp": [_P, _P, _P, _I, _P],
    "rrl_se3_exp_bwd": [_P, _P, _P, _P, _I, _P],
    "rrl_adam_gated": [_P] * 7 + [_I, _c.c_double, _c.c_double, _c.c_double, _P],
    "rrl_dense_scan": [_P, _P, _P, _P, _P, _I, _I, _I, _P],
    "rrl_fps": [_P, _P, _P, _P, _I, _I, _I, _P],
    "rrl_knn3": [_P, _P, _P, _I, _I, _I, _P],
    "rrl_sample_lines": [_P] * 8 + [_I] * 3 + [_P],
    "rrl_sample_lines_rng": [_P] * 8 + [_I] * 3 + [_P],
}
EXPORTS = sorted(list(_SIGS) + ["rrl_version", "rrl_workspace_bytes", "rrl_chamfer_workspace_bytes",
                                 "rrl_cloud_order_workspace_bytes"])

F_TARGET_KEPT = 1  # include/rrl.h RRL_F_TARGET_KEPT
F_CHAIN = 2        # RRL_F_CHAIN: leave the hit counts / CHAIN words cleared for the next step on this workspace
F_CHAINED = 4      # RRL_F_CHAINED: the previous step did (chain_left = 1): records + both scans as one launch


class ChamferRider(ctypes.Structure):
    """include/rrl.h rrl_chamfer_rider: the evaluation's Chamfer walk, carried by its culled scan's launch."""
    _fields_ = [("ws", _P), ("ws_bytes", _Z), ("best_x", _P), ("best_y", _P), ("value", _P), ("done", _c.c_int32)]


class Opts(ctypes.Structure):
    """include/rrl.h rrl_opts: the per-call options of the *_ex entry points (-1 / NULL = the library default)."""
    _fields_ = [("struct_bytes", _c.c_int32), ("flags", _c.c_int32), ("reduce_mode", _c.c_int32),
                ("deterministic", _c.c_int32), ("sort_parts", _c.c_int32), ("scan_variant", _c.c_int32),
                ("order1", _P), ("order2", _P), ("scan_counters", _P), ("scan_counter_rows", _c.c_longlong),
                ("chamfer", _P), ("payload", _P), ("problems", _c.c_int32), ("chain_left", _P)]

    def __init__(self, flags=0, reduce_mode=-1, deterministic=-1, sort_parts=-1, scan_variant=-1, order1=None,
                 order2=None, scan_counters=None, scan_counter_rows=0, chamfer=None, payload=None, problems=0,
                 chain_left=None):
        super().__init__(ctypes.sizeof(Opts), int(flags), int(reduce_mode), int(deterministic), int(sort_parts),
                         int(scan_variant), order1, order2, scan_counters, int(scan_counter_rows), chamfer, payload,
                         int(problems), chain_left)

class DemoEpochArgs(ctypes.Structure):
    """include/rrl.h rrl_demo_epoch_args (same field order)."""
    _fields_ = [("struct_bytes", _c.c_int32), ("N", _c.c_int32), ("M", _c.c_int32), ("L", _c.c_int32),
                ("rounds", _c.c_int32), ("transpose_r", _c.c_int32),
                ("rng_state", _P), ("radius", _P), ("centers", _P), ("box1", _P), ("box2", _P), ("lines", _P),
                ("filled", _P), ("tile_counts", _P),
                ("src_tri", _P), ("tar_tri", _P), ("R", _P), ("T", _P), ("ws", _P), ("ws_bytes", _Z), ("loss", _P),
                ("grad_loss", _P), ("gR", _P), ("gt", _P), ("opts", _P),
                ("cham_ws", _P), ("cham_ws_bytes", _Z), ("best_x", _P), ("best_y", _P), ("cham_value", _P),
                ("xi", _P), ("m", _P), ("v", _P), ("adam_state", _P), ("lr", _P), ("b1", _c.c_double), ("b2", _c.c_double),
                ("eps", _c.c_double), ("table", _P), ("cursor", _P), ("table_rows", _c.c_longlong), ("row", _P),
                ("pipeline", _P)]


_lib = None


class RRLError(RuntimeError):
    pass


def load():
    """Load librrl_hip.so; raises RRLError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        raise RRLError(
            f"{LIB} is missing: build it with `python __graft_entry__.py` (hipcc, gfx950). "
            "This package has no CPU or PyTorch fallback.")
    lib = ctypes.CDLL(LIB)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _I
    lib.rrl_version.restype = ctypes.c_char_p
    lib.rrl_workspace_bytes.argtypes = [_I, _I, _I, _I]
    lib.rrl_workspace_bytes.restype = _Z
    lib.rrl_chamfer_workspace_bytes.argtypes = [_I, _I, _I]
    lib.rrl_chamfer_workspace_bytes.restype = _Z
    lib.rrl_cloud_order_workspace_bytes.argtypes = [_I, _I]
    lib.rrl_cloud_order_workspace_bytes.restype = _Z
    _lib = lib
    return lib


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        raise RRLError(f"{what}: argument error {rc} (see RRL_E_* in include/rrl.h)")
    raise RRLError(f"{what}: HIP error {rc}")
