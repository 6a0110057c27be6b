"""Builds librrl_hip.so (hand-written HIP kernels + C ABI, include/rrl.h) for gfx950.

hipcc cross-compiles without a GPU.  -ffp-contract=off is REQUIRED: the label-deciding
arithmetic of the scan must not be fused.  The .so is kept in-tree
(a-robust-registration-loss_amd/lib/) so it travels to the GPU box with the snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
CSRC = os.environ.get("RRL_CSRC") or os.path.join(PKG, "csrc")  # (RRL_CSRC: A/B builds of another source tree, with RRL_HIPCC_FLAGS)
# Experimental builds (RRL_HIPCC_FLAGS = extra -D knobs of the sweep scripts) go to their OWN directory and are loaded
# only by processes that carry the same environment variable: an interrupted sweep can no longer leave a truncated
# or re-tuned library where bench.py and the tests look; the flags are also part of rrl_version().
EXP_FLAGS = os.environ.get("RRL_HIPCC_FLAGS", "").split()
LIBDIR = os.path.join(PKG, "lib_exp" if EXP_FLAGS else "lib")
LIB = os.path.join(LIBDIR, "librrl_hip.so")
SOURCES = ["rrl_scan.hip", "rrl_cull.hip", "rrl_sparse.hip", "rrl_geom.hip", "rrl_neigh.hip", "rrl_chamfer.hip", "rrl_order.hip", "rrl_epoch.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-Wall", "-Wno-unused-function"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build():
    if not os.path.exists(LIB) or EXP_FLAGS:
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(PKG), "include", "rrl.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objs, cmds = [], []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        extra = ["-fno-slp-vectorize"] if src == "rrl_cull.hip" else []  # see the file header
        extra += EXP_FLAGS  # experiments: -DNAME=value knobs
        if src == "rrl_geom.hip" and EXP_FLAGS:  # rrl_version() names them
            extra.append('-DRRL_BUILD_FLAGS="' + " ".join(EXP_FLAGS).replace('"', "'") + '"')
        cmd = [_hipcc(), *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        cmds.append(cmd)
        objs.append(obj)
    # the translation units are independent: compile them side by side (a few host cores; hipcc is single-threaded)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(cmds), max(1, (os.cpu_count() or 2) // 2))) as pool:
        list(pool.map(subprocess.check_call, cmds))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
