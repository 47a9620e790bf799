"""build.py -- compiles libmi355zk.so for gfx950 with hipcc (in-tree, next to this file).

The library is four translation units (csrc/lib_core|lib_msm|lib_ntt|lib_aux.hip) compiled in parallel and linked; a change to one
kernel family rebuilds one unit.  Staleness is decided by CONTENT (sha256 of every file a unit includes + the flags), not by mtimes, so a
copied tree (the GPU box's snapshot) never rebuilds what was built here.
"""
from __future__ import annotations

import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libmi355zk.so")
UNITS = ["lib_core", "lib_msm", "lib_ntt", "lib_aux"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DNDEBUG", "-Wno-unused-result"]
_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _closure(path: str, seen: dict) -> None:
    """every file `path` includes with quotes, transitively (the csrc tree has no conditional includes that matter)"""
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return
    with open(path, "rb") as f:
        data = f.read()
    seen[path] = data
    for inc in _INC.findall(data.decode("utf-8", "replace")):
        _closure(os.path.join(os.path.dirname(path), inc), seen)


def _unit_hash(unit: str, extra_flags) -> str:
    seen: dict = {}
    _closure(os.path.join(CSRC, unit + ".hip"), seen)
    h = hashlib.sha256(" ".join(FLAGS + list(extra_flags)).encode())
    for p in sorted(seen):
        h.update(os.path.relpath(p, HERE).encode()); h.update(seen[p])
    return h.hexdigest()


def _stale(unit: str, objdir: str, extra_flags) -> bool:
    obj = os.path.join(objdir, unit + ".o")
    try:
        with open(obj + ".hash") as f:
            return not os.path.exists(obj) or f.read().strip() != _unit_hash(unit, extra_flags)
    except OSError:
        return True


def _lib_hash(extra_flags) -> str:
    return hashlib.sha256("".join(_unit_hash(u, extra_flags) for u in UNITS).encode()).hexdigest()


def needs_build(lib: str = LIB, extra_flags=()) -> bool:
    try:
        with open(lib + ".srchash") as f:
            return not os.path.exists(lib) or f.read().strip() != _lib_hash(extra_flags)
    except OSError:
        return True


def build(force: bool = False, verbose: bool = False, lib: str = LIB, extra_flags=(), objdir: str = OBJDIR) -> str:
    """extra_flags / lib / objdir: A/B builds with compile-time switches (tools/build_variant.py) next to the shipped library"""
    extra_flags = list(extra_flags)
    if not force and not needs_build(lib, extra_flags):
        return lib
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(objdir, exist_ok=True)

    def compile_unit(unit: str) -> str:
        obj = os.path.join(objdir, unit + ".o")
        if force or _stale(unit, objdir, extra_flags):
            cmd = [hipcc] + FLAGS + extra_flags + ["-c", os.path.join(CSRC, unit + ".hip"), "-o", obj]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
            with open(obj + ".hash", "w") as f:
                f.write(_unit_hash(unit, extra_flags))
        return obj

    with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        objs = list(ex.map(compile_unit, UNITS))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl", "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(lib + ".srchash", "w") as f:
        f.write(_lib_hash(extra_flags))
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))


CPP_PROGRAMS = ("test_halo2_mirror", "test_shim_replay", "test_plonk_replay", "test_prover_process")


def build_cpp(name: str) -> str:
    """g++-compiled callers of the C-ABI under tests/cpp/ (the C++ mirror of the halo2_proofs interface, the replay of rust_shim's behaviour, and create_proof for a
    PlonkProtocol over resident buffers -- the last one is also what bench.py times as `proof_mix`).  Rebuilt when the source, the public headers or the library changed
    (content hash).  The programs link the oracle's C restatement for their own host-side CHECKS (never in a timed region): a tree without it cannot build them."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tests", "cpp", name + ".cpp")
    exe = os.path.join(root, "tests", "cpp", name)
    orc = os.path.join(root, "oracle")
    h = hashlib.sha256()
    inc = os.path.join(root, "include")
    for f in [src, os.path.join(HERE, "csrc", "fp.hpp"), os.path.join(HERE, "csrc", "slab_ranges.hpp"), os.path.join(orc, "bn254_oracle.c")] + sorted(os.path.join(inc, x) for x in os.listdir(inc)):
        with open(f, "rb") as fh:
            h.update(fh.read())
    tag = exe + ".srchash"
    try:
        fresh = os.path.exists(exe) and open(tag).read().strip() == h.hexdigest()
    except OSError:
        fresh = False
    if not fresh:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-Wno-unknown-pragmas", "-I", inc, src, "-o", exe,
                               "-L", HERE, "-lmi355zk", "-L", orc, "-loracle_bn254", f"-Wl,-rpath,{HERE}", f"-Wl,-rpath,{orc}", "-Wl,-rpath,/opt/rocm/lib"])
        with open(tag, "w") as fh:
            fh.write(h.hexdigest())
    return exe
