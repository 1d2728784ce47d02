"""Builds smpl_amd/libsmpl_amd.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsmpl_amd.so")
SOURCES = ["kernels.hip", "engine.hip", "model_compile.cpp"]
HEADERS = ["det_math.h", "device_types.h", "kernels.h", "model_compile.h", os.path.join("..", "..", "include", "smpl_amd.h")]
# -ffp-contract=off is part of the arithmetic contract (det_math.h): host and device round alike
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wall",
         "-Wno-unused-function"]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
