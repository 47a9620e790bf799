"""bench.py's launch paths, exercised without a GPU up to (not including) the device binding (VERDICT r3 missing #5 / next #5): the driver's
multi-GPU line must not be an assertion trace.  MI355_BENCH_DRYRUN=1 makes every rank print what it would bind and exit."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, extra_env=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env["MI355_BENCH_DRYRUN"] = "1"
    env.update(extra_env or {})
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    recs = [json.loads(l[l.index("{"):]) for l in out.stdout.splitlines() if '"dryrun"' in l]
    return out, recs


def test_plain_python_with_gpus_2_relaunches_itself_under_torchrun():
    out, recs = run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert out.returncode == 0, out.stderr[-800:]
    assert "re-launching" in out.stderr
    assert sorted(r["rank"] for r in recs) == [0, 1], (recs, out.stderr[-400:])
    assert all(r["world"] == 2 and r["gpus"] == 2 and r["master"] == "127.0.0.1" and r["launcher"] == "torch.distributed.run" for r in recs)
    assert sorted(r["local_rank"] for r in recs) == [0, 1]


def test_the_driver_s_torchrun_command_line():
    out, recs = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29671",
                     "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert out.returncode == 0, out.stderr[-800:]
    assert sorted(r["rank"] for r in recs) == [0, 1] and all(r["world"] == 2 for r in recs)


def test_n_1_and_single_process_do_not_relaunch():
    out, recs = run([sys.executable, "bench.py", "--gpus", "1"])
    assert out.returncode == 0 and len(recs) == 1 and recs[0]["world"] == 1 and recs[0]["launcher"] == "none"
    out, recs = run([sys.executable, "bench.py", "--gpus", "4", "--single-process"])
    assert out.returncode == 0 and len(recs) == 1 and recs[0]["single_process"] and recs[0]["world"] == 1


def test_mismatched_world_size_is_a_clear_message_not_an_assertion_trace():
    out, recs = run([sys.executable, "bench.py", "--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr and "Traceback" not in out.stderr


import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 8])
def test_ranks_share_one_srs_and_prove_one_commitment(ranks):
    """N ranks on the one visible GPU (MI355_BENCH_SHARE_GPU=1: gloo instead of RCCL, everything else the driver's command line): rank r owns points
    [r n / N, (r + 1) n / N) of ONE tau-SRS, and the folded result must be p(tau) G for the one polynomial the ranks hold together -- at N = 8, the rank count
    of the target node (BASELINE configs[4])."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    env["MI355_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1", "--master-port", str(29700 + ranks),
           "bench.py", "--gpus", str(ranks), "--logn", "20", "--steps", "2", "--warmup", "1", "--no-proof-mix", "--no-cpu-baseline", "--no-ntt", "--no-precompute"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-1500:]
    line = json.loads(next(l for l in out.stdout.splitlines() if l.startswith('{"metric"')))
    assert line["n_gpus"] == ranks and line["verified_against_field_check"] is True, line
    assert len(line["multi_gpu"]["per_rank_ms_per_step"]) == ranks
