"""ORACLE (test infrastructure). Builds the oracle's plain-C pieces with gcc into ``oracle/_build/`` (git-ignored;
travels to the GPU box with the snapshot like the product's own ``.so``) and loads them with ctypes."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
_libs = {}


def build(name: str = "jpeg_entropy", force: bool = False) -> str:
    src = os.path.join(HERE, name + ".c")
    lib = os.path.join(OUT, "lib" + name + ".so")
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        os.makedirs(OUT, exist_ok=True)
        subprocess.check_call([os.environ.get("CC", "gcc"), "-O2", "-fPIC", "-shared", "-std=c99", "-Wall", "-o", lib, src])
    return lib


def load(name: str = "jpeg_entropy") -> C.CDLL:
    if name not in _libs:
        _libs[name] = C.CDLL(build(name))
    return _libs[name]
