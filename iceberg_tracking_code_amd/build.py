"""Builds libicelk.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python -m iceberg_tracking_code_amd.build [--force]

-ffp-contract=off is part of the contract, not a tuning flag: the float sequences of the LK 2x2
solve and of the min-eigenvalue map must not be fused into FMAs or the results stop being
bit-identical to the CPU oracle (and to a non-FMA OpenCV build).
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "obj")
LIB = os.path.join(HERE, "libicelk.so")
SOURCES = ["icelk_abi.hip", "k_image.hip", "k_pyramid.hip", "k_lk.hip", "k_lk_fast.hip", "k_lk_multi.hip", "k_corners.hip", "k_corners_fast.hip", "k_sort.hip", "k_tail.hip", "k_tracks.hip", "k_utm.hip", "k_mask.hip", "k_grid.hip"]
HEADERS = [os.path.join(CSRC, "icelk_internal.h"), os.path.join(CSRC, "lk_common.h"), os.path.join(CSRC, "lk_fast_tiles.h"), os.path.join(HERE, "..", "include", "icelk.h")]
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-result", "-Wno-unused-value"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libicelk.so cannot be built (ROCm toolchain required)")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if force or _stale(obj, [path] + HEADERS):
        cmd = [_hipcc()] + FLAGS + ["-c", path, "-o", obj]
        subprocess.check_call(cmd)
    return obj


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(6, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        subprocess.check_call(cmd)
        if verbose:
            print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
