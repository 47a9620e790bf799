"""build.py -- compiles libmi355zk.so for gfx950 with hipcc (in-tree, next to this file)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmi355zk.so")
SOURCES = ["capi.hip"]
HEADERS = ["fp.cuh", "fp_asm.cuh", "fp_asm_gen.inc", "fp29.cuh", "g1.cuh", "g1_29.cuh", "msm.cuh", "ntt.cuh", "ntt29.cuh", os.path.join("..", "..", "include", "mi355zk.h")]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fgpu-rdc" if False else "-DNDEBUG",
           "-Wno-unused-result", "-o", LIB] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
