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


def test_create_proof_replay_compiles_and_fails_loudly_without_a_gpu():
    """tests/cpp/test_create_proof_replay.cpp = create_proof's step order over resident buffers, compiled: without a GPU it stops at mi355_init."""
    import torch
    exe = build_exe("test_create_proof_replay")
    if torch.cuda.is_available():
        pytest.skip("GPU present: the full replay runs under -m gpu")
    out = subprocess.run([exe, "--k", "8"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "mi355_init" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("args,env", [
    (["--layer", "4", "--k", "13"], {}),
    (["--layer", "2", "--k", "12", "--host-api"], {}),
    (["--layer", "1", "--k", "10", "--no-tables"], {}),
    (["--layer", "6", "--k", "11", "--pk-cosets", "on-the-fly"], {}),
    (["--layer", "3", "--k", "9", "--tables", "lagrange"], {}),
    (["--layer", "5", "--k", "10"], {}),
    (["--layer", "3", "--k", "10", "--pinned-witness", "--upload-threads", "3", "--early-intt", "1"], {}),
    (["--layer", "0", "--k", "9", "--advice", "70", "--fixed", "9", "--lookups", "6", "--perm", "20"], {}),
    (["--layer", "0", "--k", "8", "--advice", "12", "--fixed", "3", "--lookups", "2", "--perm", "7", "--chunk", "3", "--degree", "5", "--proofs", "3"], {}),
    (["--layer", "4", "--k", "12", "--devices", "2"], {"MI355_ALLOW_DUP_DEVICES": "1", "MI355_SHARD_MIN_LOG": "8"}),
    (["--layer", "2", "--k", "11", "--devices", "3", "--pk-cosets", "on-the-fly"], {"MI355_ALLOW_DUP_DEVICES": "1", "MI355_SHARD_MIN_LOG": "6"}),
])
def test_create_proof_replay_on_gpu(args, env):
    """create_proof_gpu_side (include/mi355zk_create_proof.hpp) for the counts of all seven layers at test sizes: polynomials
    and proving-key cosets resident (or recomputed per part), every commitment checked against p(tau) G, every evaluation against Horner, the
    quotient SEMANTICALLY (h(x) (x^n - 1) == sum_g y^g gate_g(x) from the evaluations) and both multi-open quotients with the trapdoor; two and
    three device slots: coset parts are computed on different devices by different host threads."""
    import json
    exe = build_exe("test_create_proof_replay")
    e = dict(os.environ); e.update(env)
    out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=900, env=e)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout
    line = next(l for l in out.stdout.splitlines() if l.startswith("{"))
    rec = json.loads(line)
    sh = rec["shape"]
    msm = sh["advice"] + 2 * sh["lookups"] + sh["perm_z"] + sh["quotient_pieces"] + 2
    assert rec["ok"] and rec["semantic_check"] and rec["trapdoor_check"], rec
    assert rec["msm"] == msm and rec["checked"] == msm + rec["evals"] + 1 + 2, rec
    if rec["layer"] in (2, 4) and "--k" in args and len(args) <= 5:
        assert rec["msm"] == {4: 14, 2: 11}[rec["layer"]]          # the fixtures' proof word counts
    assert rec["coset_ntt"] >= (1 + sh["advice"] + 2 * sh["lookups"] + sh["perm_z"]) * sh["quotient_pieces"]
