"""The C++ host-side mirror of the halo2_proofs interface (include/mi355zk_halo2.hpp): compiled with g++ against libmi355zk.so and the
oracle, host-only part here (domain constants vs the reference's fixture, argument checks, loud failure without a GPU); the full
program runs under -m gpu."""
import os
import subprocess

import pytest

import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_halo2_mirror")


def build_exe():
    ge.build()
    src = os.path.join(ROOT, "tests", "cpp", "test_halo2_mirror.cpp")
    pkg = os.path.join(ROOT, "scroll-prover_amd"); orc = os.path.join(ROOT, "oracle")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
                           "-L", pkg, "-lmi355zk", "-L", orc, "-loracle_bn254", f"-Wl,-rpath,{pkg}", f"-Wl,-rpath,{orc}", "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def test_cpp_mirror_host_only():
    exe = build_exe()
    out = subprocess.run([exe, "--host-only"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host-only checks passed" in out.stdout


@pytest.mark.gpu
def test_cpp_mirror_on_gpu():
    exe = build_exe()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout
