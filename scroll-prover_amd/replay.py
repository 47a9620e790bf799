"""
replay.py -- host-side driver of the compiled create_proof replay (tests/cpp/test_plonk_replay.cpp): writes a layer's PlonkProtocol (protocols.py), runs the
program as its own process and returns what it wrote -- the proof, the verifying key, the instance values (bytes in the reference's layouts) and the program's
JSON record.  Used by tests/ and bench.py, which then hand the bytes to the verifier (oracle/plonk.py, checker only); nothing here imports the oracle.
"""
from __future__ import annotations

import json
import os
import subprocess
import tempfile
import time

from . import protocols

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def exe_path() -> str:
    from . import build
    return build.build_cpp("test_plonk_replay")


def write_protocol(layer: int, directory: str, k: int | None = None, protocol_file: str | None = None, **shape) -> str:
    """protocol_file: prove a given PlonkProtocol JSON as it stands (the reference's fixtures under tests/golden/); otherwise the layer's generated protocol"""
    if protocol_file:
        return protocol_file
    path = os.path.join(directory, f"layer{layer}_protocol.json")
    with open(path, "w") as f:
        json.dump(protocols.layer_protocol(layer, k, **shape), f, separators=(",", ":"))
    return path


def run(layer: int, k: int | None = None, out_dir: str | None = None, args=(), env=None, timeout: int = 1800, protocol_file: str | None = None, **shape) -> dict:
    out_dir = out_dir or tempfile.mkdtemp(prefix=f"mi355_replay_l{layer}_")
    os.makedirs(out_dir, exist_ok=True)
    proto = write_protocol(layer, out_dir, k, protocol_file, **shape)
    e = dict(os.environ)
    e.update(env or {})
    t0 = time.perf_counter()
    out = subprocess.run([exe_path(), "--protocol", proto, "--out", out_dir] + list(args), capture_output=True, text=True, timeout=timeout, env=e)
    wall = time.perf_counter() - t0
    line = next((l for l in out.stdout.splitlines() if l.startswith("{")), None)
    rec = {"layer": layer, "ok": False, "returncode": out.returncode, "out_dir": out_dir, "protocol_path": proto, "process_wall_s": wall}
    if out.returncode != 0 or line is None:
        rec["error"] = (out.stdout + out.stderr)[-1200:]
        return rec
    rec.update(json.loads(line))
    rec["process_wall_s"] = wall
    for name in ("proof", "vk", "instances"):
        p = os.path.join(out_dir, name + ".bin")
        if os.path.exists(p):
            with open(p, "rb") as f:
                rec[name] = f.read()
    return rec


def run_process(layers, ks=None, out_dir: str | None = None, args=(), env=None, timeout: int = 2400, shapes=None) -> dict:
    """tests/cpp/test_prover_process.cpp: ONE process holding the SRS, proving keys and witnesses of several layers (a chunk prover {0, 1, 2}, a batch prover {3, 4}), under the HBM
    plan of plan_residency, proofs back to back.  Returns the program's record with, per layer, the bytes it wrote."""
    from . import build
    out_dir = out_dir or tempfile.mkdtemp(prefix="mi355_prover_process_")
    os.makedirs(out_dir, exist_ok=True)
    protos = []
    for i, layer in enumerate(layers):
        fx = os.path.join(ROOT, "tests", "golden", f"protocol_layer{layer}.json")
        k = (ks or {}).get(layer)
        protos.append(fx if (os.path.exists(fx) and not k) else write_protocol(layer, out_dir, k, None, **((shapes or {}).get(layer) or {})))
    cmd = [build.build_cpp("test_prover_process"), "--out", out_dir] + [x for p in protos for x in ("--protocol", p)] + list(args)
    e = dict(os.environ); e.update(env or {})
    t0 = time.perf_counter()
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=e)
    line = next((l for l in out.stdout.splitlines() if l.startswith("{")), None)
    rec = {"ok": False, "returncode": out.returncode, "out_dir": out_dir, "protocol_paths": protos, "process_wall_s": time.perf_counter() - t0}
    if out.returncode != 0 or line is None:
        rec["error"] = (out.stdout + out.stderr)[-1200:]
        return rec
    rec.update(json.loads(line))
    for i, lay in enumerate(rec["layers"]):
        for name in ("proof", "vk", "instances"):
            with open(os.path.join(out_dir, str(i), name + ".bin"), "rb") as f:
                lay[name] = f.read()
        lay["protocol_path"] = protos[i]
    return rec
