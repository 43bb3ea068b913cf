"""Build the HIP shared library in-tree with hipcc for gfx950.

``python -m playaid_core_amd._build`` or ``build()``; the resulting
``playaid_core_amd/libplayaid_hip.so`` is git-ignored but travels to the GPU
box with the snapshot. hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libplayaid_hip.so")
ARCH = "gfx950"

# (source, extra flags). preprocess.hip reproduces Pillow/OpenCV arithmetic
# bit-for-bit and must not have its multiplies and adds fused.
SOURCES = [
    ("igemm.hip", []),
    ("pigemm.hip", []),
    ("psgemm.hip", []),
    ("igemm_bf16.hip", []),
    ("patchconv.hip", []),
    ("patchconv_bf16.hip", []),
    ("wino.hip", []),
    ("stem.hip", []),
    ("stem_pool.hip", []),
    ("misc.hip", []),
    ("preprocess.hip", ["-ffp-contract=off"]),
    ("detect.hip", ["-ffp-contract=off"]),
    ("lstm.hip", []),
    ("convnet.hip", []),
    ("transformer.hip", []),
    ("jpeg.hip", ["-ffp-contract=off"]),
    ("mjpeg.hip", []),
    ("savebox.hip", ["-ffp-contract=off"]),
    ("yolo.hip", ["-ffp-contract=off"]),
    ("pa_api.hip", []),
]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "hipcc")
    headers = [os.path.join(CSRC, "pa_kernels.h"), os.path.join(CSRC, "jpeg_dct.h"), os.path.join(HERE, "..", "include", "playaid_hip.h")]
    objs = []
    rebuilt = False
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not _newer(o, [s] + headers + [os.path.abspath(__file__)]):
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-c", s, "-o", o] + extra
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            rebuilt = True
    # (an object compiled by hand is newer than the library without this call having rebuilt anything: link then too)
    if rebuilt or force or not _newer(LIB, objs):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
