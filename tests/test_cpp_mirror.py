"""The C++ host-side mirror of the halo2_proofs interface (include/mi355zk_halo2.hpp): compiled with g++ against libmi355zk.so and the
oracle, host-only part here (domain constants vs the reference's fixture, argument checks, loud failure without a GPU); the full
program runs under -m gpu."""
import os
import subprocess

import pytest

import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_halo2_mirror")


def build_exe(name="test_halo2_mirror"):
    ge.build()
    return ge.build_cpp(name)


def protocol_dir(tmp_path):
    """all seven layers' PlonkProtocols at full size (scroll-prover_amd/protocols.py; layers 2 / 4 equal the reference's fixtures, tests/test_plonk_protocol.py)"""
    zk = ge.load_package()
    for layer in range(7):
        zk.protocols.write(layer, str(tmp_path / f"layer{layer}.json"))
    return str(tmp_path)


def test_cpp_mirror_host_only(tmp_path):
    """domain constants vs the reference's fixture, the compressed-point codec, and the plan compiler / residency rule of create_proof on all seven protocols"""
    exe = build_exe()
    out = subprocess.run([exe, "--host-only", "--protocols", protocol_dir(tmp_path)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host-only checks passed" in out.stdout


@pytest.mark.gpu
def test_cpp_mirror_on_gpu():
    exe = build_exe()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout


def test_shim_replay_compiles_and_fails_loudly_without_a_gpu():
    """tests/cpp/test_shim_replay.cpp = what rust_shim/mi355zk.rs does, compiled: here it must build and stop at mi355_init (exit 2)."""
    import torch
    exe = build_exe("test_shim_replay")
    if torch.cuda.is_available():
        pytest.skip("GPU present: the full replay runs under -m gpu")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "mi355_init" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_shim_replay_on_gpu():
    """register by struct-owned handles, clone + downsize as load_params_map does, commits on sub-slices, freed-and-reused addresses,
    8 concurrent committing threads with per-thread options, release in any order."""
    exe = build_exe("test_shim_replay")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout
