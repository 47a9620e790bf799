#!/usr/bin/env python3
"""
bench.py -- the headline measurement of BASELINE.json: BN254 G1 MSM (G1-adds/s) + Fr NTT (butterflies/s) at k = 26
on MI355X, one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input: ONE commitment = one G1 MSM over 2^k
(scalar, point) pairs (ParamsKZG::commit / best_multiexp; create_proof issues 11-14 of these per compression-layer
proof).  Inputs (scalars and the SRS basis) are resident in HBM when the timed region starts; the
result (96 B) comes back to the host inside it, exactly as the C-ABI delivers it.
N > 1: the 2^k pairs are sharded by point range: rank r owns 2^k / N points + scalars, computes its
partial sum, the 96-byte partials are all-gathered with RCCL and folded on the device ("scaling": "strong").
The NTT (fwd + inv, natural order, ifft divisor included) is timed in the same run as a secondary figure; it does
not shard at k <= 26 (replicas only), so each rank runs its own copy.

Synthetic data: basis g[i] = tau_r^i * G built on the device (ParamsKZG::setup restated, so every commitment can be
checked in the field: commit(p) = p(tau) G); scalars uniform in [0, r) from a seeded generator.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

R_TOP = 0x30644E72E131A029
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# ALU-side ceiling of the dominant kernel: the 29-bit XYZZ mixed-addition chain of tools/microbench.hip at this kernel's occupancy
# (3 waves/SIMD, operands cache-resident), profiles/r01_microbench_final.log.  Reported next to the HBM roofline, never instead of it.
MADD_CHAIN_PEAK = 15.9e9
MADS_PER_MADD = 1467   # v_mad_[ui]64 on the common path of one XYZZ mixed addition (tools/isa_block_census.py)


def source_hash(kind: str) -> str:
    """sha256[:16] of the kernel sources a recorded counter figure belongs to (profiles/pmc_latest.json stores it; a figure whose sources changed is not re-emitted)"""
    import hashlib
    files = {"msm": ("msm.hpp", "lib_msm.hip", "fp29.hpp", "g1_29.hpp"), "ntt": ("ntt29.hpp", "lib_ntt.hip", "fp29.hpp")}[kind]
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, "scroll-prover_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def recorded_traffic(key: str, kind: str):
    """HBM bytes from the separate rocprofv3 --pmc passes (tools/collect_profiles.sh + tools/pmc_report.py): counter collection cannot share a run with the timing, so the
    figure is RECORDED -- and only re-emitted while the kernel sources it was measured on are the ones that just ran"""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    except Exception:
        return None, None
    want = (rec.get("source_sha16") or {}).get(kind)
    if want != source_hash(kind):
        return None, f"stale: {rec.get('source')} was measured on other {kind} sources (recorded {want}, now {source_hash(kind)}); re-run tools/collect_profiles.sh"
    return rec.get(key), rec.get("source")


LINE_BUDGET = 4096   # bytes of the ONE stdout line (the driver keeps a bounded tail of the output: round 5's 25 KB line was not parsed)


def _clean(x, nd=6):
    """strict-JSON value: no NaN / Infinity (-> null), floats to `nd` significant digits, numpy scalars to Python ones"""
    import math
    if isinstance(x, dict):
        return {str(k_): _clean(v, nd) for k_, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_clean(v, nd) for v in x]
    if isinstance(x, (bool, type(None), str, int)):
        return x
    if isinstance(x, np.generic):
        x = x.item()
        if isinstance(x, (bool, int, str)):
            return x
    if isinstance(x, float):
        if not math.isfinite(x):
            return None
        if x.is_integer() and abs(x) < 2.0 ** 53:
            return int(x)   # counts that travelled as floats (pairs per launch, bytes) keep every digit
        return float(f"{x:.{nd}g}")
    return str(x)


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 3] + "..."


def compact_line(full: dict, detail_file: str | None) -> str:
    """The ONE line the driver reads: the contract's keys, `config` (workload + the other two legs of BASELINE.json's metric), `roofline`, `cpu_baseline` -- at most
    LINE_BUDGET bytes of strict JSON whatever the run produced.  Everything else (proof mix, sizes, batched commitments, witness-like column, host route, phase
    breakdowns) is in `detail_file`.  tests/test_bench_line.py holds a worst-case record against the budget."""
    cfg = full.get("config") or {}
    keep_cfg = ("workload", "log_n", "window_bits", "windows", "parallelism", "srs_window_tables", "ntt_k26_ms", "ntt_k26_butterflies_per_s", "ntt_roofline_frac_hbm",
                "proxies_ms", "proxies_shape", "proxies_separate_processes_ms", "layer_ms", "chunk_prover_process_ms", "chunk_prover_process_peak_hbm_gib",
                "verifier_accepts_released_reference_proofs", "all_checks")
    c = {k_: cfg[k_] for k_ in keep_cfg if k_ in cfg}
    for k_ in list(cfg):   # the NTT keys carry the size in their name when --logn is not 26
        if k_.startswith("ntt_k") and k_ not in c:
            c[k_] = cfg[k_]
    for k_ in ("workload", "parallelism", "proxies_shape"):
        if k_ in c:
            c[k_] = _short(c[k_], 200)
    rf = full.get("roofline") or {}
    r = {k_: rf.get(k_) for k_ in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "pairs_per_launch")}
    r["traffic_source"] = _short(rf.get("traffic_source"), 120) if rf.get("traffic_source") else None
    alu = rf.get("alu") or {}
    r["alu"] = {k_: alu.get(k_) for k_ in ("frac", "achieved", "peak", "unit")}
    cb = full.get("cpu_baseline")
    if cb is not None:
        cb = {k_: cb.get(k_) for k_ in ("value", "unit", "cores", "kind", "sample", "pairs_per_s")}
        cb["sample"] = _short(cb.get("sample"), 200)
    out = {k_: full.get(k_) for k_ in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    out["dtype"] = _short(out["dtype"], 80)
    out.update({"config": c, "roofline": r, "cpu_baseline": cb, "pairs_per_s": full.get("pairs_per_s"), "verified_against_field_check": full.get("verified_against_field_check")})
    mg = full.get("multi_gpu")
    if mg is not None:
        m = {k_: mg.get(k_) for k_ in ("distinct_devices", "backend", "rccl_version", "mode", "exchange") if k_ in mg}
        if "exchange" in m:
            m["exchange"] = _short(m["exchange"], 100)
        if isinstance(mg.get("devices"), int):
            m["devices"] = mg["devices"]
        if mg.get("per_rank_ms_per_step"):
            m["per_rank_ms_per_step"] = [round(float(x), 2) for x in mg["per_rank_ms_per_step"]][:16]
        out["multi_gpu"] = m
    out["detail_file"] = detail_file
    out = _clean(out)
    line = json.dumps(out, allow_nan=False, separators=(", ", ": "))
    # whatever the run produced, the line stays inside the budget: drop the optional keys first, last the long strings
    for victim in (("config", "layer_ms"), ("multi_gpu",), ("roofline", "alu"), ("config", "proxies_shape"), ("roofline", "traffic_source"), ("cpu_baseline", "sample"), ("config", "parallelism")):
        if len(line.encode()) <= LINE_BUDGET:
            break
        d = out
        for k_ in victim[:-1]:
            d = d.get(k_) or {}
        d.pop(victim[-1], None)
        line = json.dumps(out, allow_nan=False, separators=(", ", ": "))
    assert len(line.encode()) <= LINE_BUDGET, len(line)
    return line


def emit(full: dict) -> None:
    """full record -> bench_detail.json (repo root, and gpurun_out/ when that directory exists so that it travels back from a GPU box); compact line -> stdout, last"""
    full = _clean(full, nd=9)
    detail_file = None
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if not os.path.isdir(d):
            continue
        try:
            with open(os.path.join(d, "bench_detail.json"), "w") as f:
                json.dump(full, f, allow_nan=False, indent=1)
            detail_file = detail_file or os.path.relpath(os.path.join(d, "bench_detail.json"), ROOT)
        except OSError as e:   # a read-only tree must not cost the line
            sys.stderr.write(f"bench.py: could not write {d}/bench_detail.json: {e}\n")
    sys.stderr.write(f"bench.py: full record ({len(json.dumps(full))} bytes) in {detail_file}; the line below is its summary\n")
    sys.stderr.flush()
    print(compact_line(full, detail_file), flush=True)


def rand_scalars(n: int, seed: int, device) -> torch.Tensor:
    """n field elements as [n,4] int64 limbs (Montgomery form of uniformly random elements), generated on the device."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    a = torch.randint(0, 256, (n * 32,), dtype=torch.uint8, device=device, generator=gen).view(torch.int64).view(n, 4)
    top = a[:, 3] & ((1 << 62) - 1)
    top = torch.where(top >= R_TOP, top >> 1, top)  # keep the 254-bit value below r
    a[:, 3] = top
    return a


def witness_like_scalars(n: int, seed: int, device, h2) -> torch.Tensor:
    """second distribution of SURVEY 8d: 60 % zero, 20 % in 1..255, 10 % 64-bit values, 10 % uniform (selector / byte / lookup columns).
    Small values are looked up in tables of Montgomery forms (x * R mod r is not a bit pattern torch can synthesise); built in chunks."""
    gen = torch.Generator(device=device); gen.manual_seed(seed)
    a = rand_scalars(n, seed + 1, device)
    small = torch.from_numpy(np.stack([h2.fr(v) for v in range(256)]).view(np.int64)).to(device)
    mid_tab = torch.from_numpy(np.stack([h2.fr((v * 0x9E3779B97F4A7C15) & ((1 << 64) - 1)) for v in range(4096)]).view(np.int64)).to(device)
    step = 1 << 22
    for lo in range(0, n, step):
        m = min(step, n - lo)
        u = torch.rand(m, device=device, generator=gen)
        pick = torch.randint(1, 256, (m,), device=device, generator=gen)
        pick_mid = torch.randint(0, 4096, (m,), device=device, generator=gen)
        blk = a[lo:lo + m]
        blk = torch.where((u < 0.9).unsqueeze(1), torch.index_select(mid_tab, 0, pick_mid), blk)
        blk = torch.where((u < 0.8).unsqueeze(1), torch.index_select(small, 0, pick), blk)
        blk = torch.where((u < 0.6).unsqueeze(1), torch.zeros_like(blk), blk)
        a[lo:lo + m] = blk
    return a.contiguous()


def released_proofs_check():
    """CHECKER leg (oracle/ is test infrastructure; nothing here is timed or shipped): the verifier that judges this run's proofs (oracle/plonk.py) on the proofs the REFERENCE released --
    the ten committed under tests/golden/ (7 chunk proofs, 2 batch proofs under the Poseidon transcript; the bundle proof under Keccak with the generated layer-6 protocol) --, each under
    a real pairing check.  Returns how many were accepted, or None if the check could not run (it must never take the bench line down)."""
    try:
        from oracle import plonk, pyref
        import importlib.util as ilu
        spec = ilu.spec_from_file_location("_protocols", os.path.join(ROOT, "scroll-prover_amd", "protocols.py")); protocols = ilu.module_from_spec(spec); spec.loader.exec_module(protocols)
        gold = os.path.join(ROOT, "tests", "golden")
        kat = json.load(open(os.path.join(gold, "kat.json")))
        neg = pyref.g2_from_evm_words([int(w, 16) for w in kat["yul"]["s_g2_words"]])
        words = lambda b: [int.from_bytes(b[i:i + 32], "big") for i in range(0, len(b), 32)]
        l2 = plonk.Protocol(json.load(open(os.path.join(gold, "protocol_layer2.json")))); l4 = plonk.Protocol(json.load(open(os.path.join(gold, "protocol_layer4.json"))))
        cases = [(l2, kat["chunk_proof"])] + [(l2, m) for m in kat["more_chunk_proofs"]] + [(l4, kat["batch_proof"]), (l4, kat["batch_proof_2"])]
        n = sum(bool(plonk.verify(pr, None, words(bytes.fromhex(c["instances"])), bytes.fromhex(c["proof"]), transcript="poseidon", neg_s_g2=neg)["ok"]) for pr, c in cases)
        pd, pi, vk = bytes.fromhex(kat["bundle_proof_data"]), bytes.fromhex(kat["bundle_pi_data"]), bytes.fromhex(kat["vk_bundle"])
        n += bool(plonk.verify(plonk.Protocol(protocols.layer_protocol(6)), None, words(pd[:384]) + words(pi), pd[384:], transcript="evm", neg_s_g2=neg,
                               preprocessed=[pyref.g1_decompress(vk[8 + 32 * i:8 + 32 * i + 32]) for i in range(7)], initial_state=int(kat["yul"]["transcript_initial_state"]))["ok"])
        return int(n), len(cases) + 1
    except Exception as e:   # noqa: BLE001 -- a checker problem is reported, not raised
        sys.stderr.write(f"released_proofs_check could not run: {e!r}\n")
        return None


def replay_create_proof(layer: int, k: int | None = None, host_api: bool = False, timeout: int = 1500, devices: int = 1, **shape):
    """One layer of scroll-prover's proof stack, proven on the device from the layer's PlonkProtocol by the compiled caller tests/cpp/test_plonk_replay.cpp
    (mi355zk::plonk::create_proof, include/mi355zk_plonk.hpp; its own process, run BEFORE this process binds the GPU), then VERIFIED here from the bytes it
    wrote: oracle/plonk.py (checker only) re-derives every challenge, walks the protocol's JSON expression tree for h(x) (x^n - 1) == numerator(x) and checks
    the SHPLONK opening with the synthetic SRS's trapdoor.  Layers 2 and 4 run the reference's OWN protocol files (tests/golden/protocol_layer{2,4}.json =
    [REF release-v0.13.1/chunk.protocol], the `protocol` of [REF integration/tests/test_data/full_proof_batch_agg_1.json])."""
    zk = ge.load_package()
    fx = os.path.join(ROOT, "tests", "golden", f"protocol_layer{layer}.json")
    args = (["--host-api"] if host_api else []) + (["--devices", str(devices)] if devices > 1 else [])
    rec = zk.replay.run(layer, k, args=args, timeout=timeout, protocol_file=fx if (os.path.exists(fx) and not k and not shape) else None, **shape)
    if not rec.get("ok"):
        return {"layer": layer, "ok": False, "error": rec.get("error")}
    from oracle import plonk
    t0 = time.perf_counter()
    pr = plonk.Protocol(json.load(open(rec["protocol_path"])))
    inst = plonk.mont_to_ints(np.frombuffer(rec["instances"], dtype=np.uint64).reshape(-1, 4))
    try:
        ver = plonk.verify(pr, rec["vk"], inst, rec["proof"], 0x5343524F4C4C0001 + (rec["layer"] if rec["layer"] >= 0 else 0), transcript=rec["transcript"])
    except AssertionError as e:
        ver = {"ok": False, "error": str(e)}
    out = {k_: v for k_, v in rec.items() if k_ not in ("proof", "vk", "instances", "out_dir", "protocol_path", "returncode")}
    out.update({"layer": layer, "verified": bool(ver["ok"]), "verify_s": round(time.perf_counter() - t0, 2), "ok": bool(ver["ok"]),
                "protocol_source": "reference fixture" if rec["protocol_path"].startswith(os.path.join(ROOT, "tests", "golden")) else pr.d.get("source")})
    import shutil
    shutil.rmtree(rec["out_dir"], ignore_errors=True)
    return out


def prover_process(layers, timeout: int = 2400):
    """ONE process holding the SRS, proving keys and witnesses of several layers under plan_residency (tests/cpp/test_prover_process.cpp), their proofs back to back, each
    verified here from its bytes: the shape of a ChunkProver / BatchProver process [REF integration/src/prove.rs:11-21,30-43]"""
    zk = ge.load_package()
    from oracle import plonk
    rec = zk.replay.run_process(list(layers), timeout=timeout)
    if not rec.get("ok"):
        return {"ok": False, "layers": list(layers), "error": rec.get("error")}
    ok = True
    for lay in rec["layers"]:
        pr = plonk.Protocol(json.load(open(lay["protocol_path"])))
        inst = plonk.mont_to_ints(np.frombuffer(lay["instances"], dtype=np.uint64).reshape(-1, 4))
        try:
            lay["verified"] = bool(plonk.verify(pr, lay["vk"], inst, lay["proof"], int(lay["tau"], 16), transcript=lay["transcript"])["ok"])
        except AssertionError:
            lay["verified"] = False
        ok = ok and lay["verified"]
        for key in ("proof", "vk", "instances", "protocol_path"):
            lay.pop(key, None)
    import shutil
    shutil.rmtree(rec["out_dir"], ignore_errors=True)
    out = {k_: v for k_, v in rec.items() if k_ not in ("out_dir", "protocol_paths", "returncode")}
    out["ok"] = out["every_proof_verified"] = ok
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--logn", type=int, default=26, help="log2 of the MSM / NTT size (BASELINE metric: 26)")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--table-bits", type=int, default=0, help="window bits of the registration-time tables (mi355_srs_precompute; 0 = automatic, up to 24)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-log", type=int, default=26, help="log2 of the sample the CPU baselines run on (default: the full workload, ~25 s MSM + ~10 s NTT on 16 CPUs)")
    ap.add_argument("--no-batch-legs", action="store_true", help="skip the many-column legs (110 x 2^21, 256 x 2^20 through mi355_msm_g1_batch_dev)")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--no-precompute", action="store_true", help="skip the registration-time window tables (mi355_srs_precompute)")
    ap.add_argument("--no-proof-mix", action="store_true", help="skip the compiled create_proof replays (all seven layers from their PlonkProtocols, each proof verified from its bytes)")
    ap.add_argument("--proxy-chunk-proof", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--prover-process", action="store_true", help=argparse.SUPPRESS)   # accepted for older command lines: on by default since round 6
    ap.add_argument("--no-prover-process", action="store_true", help="skip the chunk prover as ONE process (layers 0 + 1 + 2 in tests/cpp/test_prover_process.cpp: three SRS degrees, three proving keys under plan_residency, proofs back to back; ~80 s).  Without it config.proxies_ms[0] falls back to the sum of three single-layer processes and says so")
    ap.add_argument("--single-process", action="store_true", help="N GPUs behind ONE process (mi355_init_multi: shards, worker threads, ncclAllGather inside the library) instead of one rank per GPU")
    ap.add_argument("--no-host-api", action="store_true", help="skip the host-pointer leg (mi355_msm_g1_host: scalars cross PCIe inside the call; reported next to, never as, the headline value)")
    ap.add_argument("--no-table-free", action="store_true", help="skip the leg without window tables")
    ap.add_argument("--no-witness-like", action="store_true", help="skip the witness-like column (with every other leg off the run then launches only the timed uniform MSMs: the profile whose per-kernel averages are comparable with roofline.avg_launch_ms)")
    ap.add_argument("--no-sizes", action="store_true", help="skip the k = 20 / 24 legs (MSM on prefix views of the basis + NTT)")
    ap.add_argument("--host-api", action="store_true", help=argparse.SUPPRESS)   # accepted for older command lines: the leg is on by default now
    args = ap.parse_args()

    single = args.single_process          # N GPUs behind this one process (the reference's shape: one prover process, one params_map)
    if args.gpus > 1 and not single and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher (the driver starts N = 1 exactly like this): become the launch the contract describes --
        # one rank per GPU under torch.distributed.run on 127.0.0.1 -- instead of failing on WORLD_SIZE (VERDICT r3 missing #5 / next #5)
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print(f"bench.py: --gpus {args.gpus} without WORLD_SIZE, re-launching as: {' '.join(cmd)}", file=sys.stderr, flush=True)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if single and world > 1:
        raise SystemExit("bench.py: --single-process drives all GPUs from ONE process; launch it without torchrun")
    if not single and world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}; pass --gpus {world} (or --single-process)")
    if os.environ.get("MI355_BENCH_DRYRUN") == "1":
        # launch-path check for GPU-less boxes (tests/test_bench_launch.py): everything up to, not including, the device binding
        print(json.dumps({"dryrun": True, "rank": rank, "world": world, "local_rank": local_rank, "gpus": args.gpus, "single_process": single,
                          "master": os.environ.get("MASTER_ADDR"), "launcher": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "none"}), flush=True)
        return
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    if world > 1 and os.environ.get("MI355_BENCH_SHARE_GPU") != "1" and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    # one rank per GPU.  MI355_BENCH_SHARE_GPU=1 (test only) lets several ranks share the visible GPUs and moves the
    # collective to gloo, so the N > 1 control flow can be exercised on a one-GPU box; the driver never sets it.
    share = os.environ.get("MI355_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)   # "nccl" == RCCL on ROCm

    # ---- "chunk-proof wall-clock" (third part of the BASELINE metric): not producible here (no Rust / Go / SRS / trace).  The stand-in: every layer of the proof
    # stack [REF integration/src/prove.rs:36-43,67,95-97] (chunk proof = layers 0 + 1 + 2, batch proof = 3 + 4, bundle = 5 + 6) is PROVEN on the device from the
    # layer's PlonkProtocol -- layers 2 and 4 from the reference's own protocol files: 11 / 14 commitments, 5 / 8 inverse transforms, 20 / 32 coset transforms,
    # 17 / 27 evaluations, 896 / 1 312 proof bytes -- by mi355zk::plonk::create_proof (include/mi355zk_plonk.hpp), and each proof is VERIFIED from its bytes.
    # Each replay is its own process and runs before this one binds the GPU; two proofs per process, the second (steady-state) one is reported;
    # resident = polynomials stay in HBM between the C-ABI calls, host_api = every operand crosses PCIe.
    proof_mix = None
    if world == 1 and not single and rank == 0 and not args.no_proof_mix and args.logn == 26:
        host_layers = () if args.no_host_api else (0, 2, 4)   # the host-pointer route (every operand crosses PCIe both ways) next to the resident one: the reference's two protocols and the many-column layer
        L = {}
        for lay in (4, 6, 2, 1, 3, 5, 0):
            L[lay] = replay_create_proof(lay, host_api=lay in host_layers)
        ok_all = all(r.get("ok") for r in L.values())
        l4 = L[4]
        hb = None
        if l4.get("host_api_ms", -1) > 0 and l4.get("host_api_fft_batched_ms", -1) > 0:   # the host route with the transform loops through mi355_ntt_fr_batch_host
            hb = round(l4["host_api_ms"] - l4["host_api_fft_ms"] + l4["host_api_fft_batched_ms"], 3)

        def proxy(what, layers):
            rs = [L[x] for x in layers]
            good = all(r.get("ok") for r in rs)
            host = [r.get("host_api_ms", -1) for r in rs]
            return {"what": what, "resident_ms": sum(r.get("resident_ms", 0) for r in rs) if good else None,
                    "first_proof_ms": sum(r.get("first_proof_ms", 0) for r in rs) if good else None,
                    "host_api_ms": sum(host) if good and all(h is not None and h > 0 for h in host) else None,
                    "commitments": sum(r.get("msm", 0) for r in rs), "peak_hbm_gib": {f"layer{x}": (L[x].get("hbm") or {}).get("peak_used_gib") for x in layers},
                    "every_proof_verified": all(r.get("verified") for r in rs),
                    **{f"layer{x}": L[x] for x in layers}}
        proof_mix = {"c_abi_resident_ms": l4.get("resident_ms"), "host_api_ms": l4.get("host_api_ms"), "host_api_batched_fft_ms": hb, "layer4": l4,
                     "chunk_proof_proxy": proxy("layer 0 (k = 20 inner circuit: a stated-shape stand-in, 800 advice / 60 lookups / 150 permuted columns / degree 9) + layer 1 (k = 24, halo2-base rule on layer1.config) + layer 2 (k = 25, the reference's chunk.protocol): the three create_proof calls of gen_halo2_chunk_proof", (0, 1, 2)),
                     "batch_proof_proxy": proxy("layer 3 (k = 21, halo2-base rule on layer3.config: 93 advice, 8 lookups, 32 grand products) + layer 4 (k = 26, the reference's batch protocol): gen_batch_proof", (3, 4)),
                     "bundle_proof_proxy": proxy("layer 5 (k = 21) + layer 6 (k = 26, layer 2's constraint system at the bundle's degree): gen_bundle_proof", (5, 6)),
                     "every_proof_verified": ok_all,
                     # optional (--prover-process): the chunk prover as ONE process -- the three layers' SRS, proving keys and witnesses resident together under the HBM plan
                     "chunk_prover_process": prover_process((0, 1, 2)) if not args.no_prover_process else None,
                     "excludes": "witness synthesis of the real circuits (the circuit instances are built to satisfy each protocol; layers 0-5 run the reference's Poseidon transcript, layer 6 its Keccak transcript and EVM proof layout)"}
    if single and args.gpus > 1 and not args.no_proof_mix and args.logn == 26:
        # N devices behind ONE prover process: witness columns live round-robin on the devices, commitments take scalars from whichever device
        # holds them (shards of the basis everywhere), the iNTT batch and the coset parts of the quotient run concurrently on different devices
        # (per-device locks, one host thread per device) -- the layer-4 mix with the other GPUs given work during the NTT phase
        l4m = replay_create_proof(4, host_api=False, devices=args.gpus)
        proof_mix = {"c_abi_resident_ms": l4m.get("resident_ms"), "devices": args.gpus, "layer4": l4m}
    zk = ge.load_package()
    lib, check, ptr, h2 = zk._capi.lib(), zk._capi.check, zk._capi.ptr, zk.halo2
    if single:
        # device ids 0..N-1; on a box with fewer GPUs the same device may be listed twice when MI355_ALLOW_DUP_DEVICES=1 (test mode)
        cnt = torch.cuda.device_count()
        zk.init([i % cnt for i in range(args.gpus)] if os.environ.get("MI355_ALLOW_DUP_DEVICES") == "1" else list(range(args.gpus)))
    else:
        zk.init(dev_index)
    check(lib.mi355_set_stream(C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    check(lib.mi355_msm_set_window_bits(args.window_bits))

    k = args.logn
    n_total = 1 << k
    assert n_total % world == 0
    n = n_total // world                      # pairs owned by this rank (point-range shard)

    # ---- synthetic inputs, resident in HBM.  ONE tau for the whole job: rank r owns the points g[r n .. (r + 1) n) = tau^(r n + i) G of the SAME SRS, so that the
    # N > 1 result is the one identity a sharded commitment must satisfy, commit(p) = p(tau) G over the whole 2^k-point basis
    tau = 0x5343524F4C4C0001                  # seed ("SCROLL", 1)
    shard_k = (n - 1).bit_length()
    g = torch.empty(n * 64, dtype=torch.uint8, device=dev)
    tau_m = h2.fr(tau)
    if world == 1:
        gl_scratch = torch.empty(n * 64, dtype=torch.uint8, device=dev)
        w_shard = pow(h2.FR_ROOT_OF_UNITY, 1 << (h2.FR_S - shard_k), h2.R_MOD)
        check(lib.mi355_srs_setup_dev(ptr(g), ptr(gl_scratch), shard_k, ptr(tau_m), ptr(h2.fr(w_shard))))
        del gl_scratch
    else:
        # this rank's slice only: scalars tau^(r n) * tau^i by a constant fill and distribute_powers, then the fixed-base multiples (what mi355_srs_setup_dev does for i < n)
        pw = torch.empty((n, 4), dtype=torch.int64, device=dev)
        h2.gate_eval(pw, [], [(h2.fr(pow(tau, rank * n, h2.R_MOD)), [])], n)
        check(lib.mi355_distribute_powers_fr_dev(ptr(pw), n, ptr(tau_m)))
        check(lib.mi355_g1_fixed_base_mul_dev(ptr(g), ptr(pw), n))
        check(lib.mi355_synchronize())
        del pw
    handle = C.c_uint64()
    check(lib.mi355_srs_register_dev(ptr(g), n, 0, C.byref(handle)))
    pre_ms = None
    if not args.no_precompute and not args.window_bits:
        # registration-time work, outside the timed region: T[w][i] = 2^(c w) P_i (W x the basis in HBM: 48 GiB at 2^26, c = 22)
        tp = time.perf_counter()
        check(lib.mi355_srs_precompute(handle.value, 0, args.table_bits))
        pre_ms = (time.perf_counter() - tp) * 1e3
    scalars = rand_scalars(n, 0x5343524F4C4C0002 + rank, dev)
    torch.cuda.synchronize()

    out = np.zeros(12, dtype=np.uint64)

    def local_msm():
        check(lib.mi355_msm_g1_dev(handle.value, 0, ptr(scalars), n, ptr(out)))
        return out

    def step():
        if world == 1:
            return local_msm()
        # RCCL over xGMI: one all_gather of 96 B per rank, then the fold on the device (scroll-prover_amd/distributed.py)
        if share:
            return zk.distributed.sharded_multiexp(local_msm, h2.g1_sum, "cpu")
        return zk.distributed.sharded_multiexp_device(zk._capi, handle.value, scalars, n)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        result = step()
    check(lib.mi355_profile_reset())
    check(lib.mi355_profile_enable(1))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = step()
    barrier()
    dt = time.perf_counter() - t0
    check(lib.mi355_profile_enable(0))
    result = np.array(result, copy=True)   # `out` is reused by every later leg: keep the timed steps' result for the checks below
    multi = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if share else dev)
        per_rank = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(per_rank, t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # what actually ran, so that a silent single-device fallback cannot hide in an N > 1 line: every rank's own device and time
        names = [None] * world
        dist.all_gather_object(names, {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "name": torch.cuda.get_device_name(dev_index),
                                       "pci_bus_id": getattr(torch.cuda.get_device_properties(dev_index), "pci_bus_id", None), "uuid": str(getattr(torch.cuda.get_device_properties(dev_index), "uuid", ""))})
        try:
            rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            rccl_version = None
        multi = {"devices": names, "distinct_devices": len({(d["device_index"], d["uuid"]) for d in names}),
                 "exchange": "gloo all_gather + host fold (MI355_BENCH_SHARE_GPU test mode)" if share else "RCCL all_gather_into_tensor of 96-B partials + device fold (mi355_g1_sum_dev)",
                 "backend": "gloo" if share else "nccl (RCCL)", "rccl_version": rccl_version, "per_rank_ms_per_step": [float(x.item()) / args.steps * 1e3 for x in per_rank]}

    def prof(name):
        ms, cnt = C.c_double(), C.c_uint64()
        check(lib.mi355_profile_get(name.encode(), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    c_, w_, e_ = C.c_int(), C.c_int(), C.c_uint64()
    check(lib.mi355_msm_last_plan(C.byref(c_), C.byref(w_), C.byref(e_)))
    c, W = c_.value, w_.value
    acc_ms, acc_cnt = prof("msm_accumulate")
    # per MSM (summed over the chunks of a pipelined MSM: the sort of chunk k + 1 runs under the accumulation of chunk k, so the
    # phases overlap and do not add up to msm_total)
    n_msm = max(1, prof("msm_total")[1])
    phases = {p: prof(p)[0] / n_msm for p in ("msm_digits", "msm_sort", "sort_l1", "sort_hist", "sort_l2", "msm_accumulate", "msm_reduce", "msm_total")}
    chunks = max(1, round(acc_cnt / n_msm))

    # ---- correctness of what was timed: commit(p) = p(tau) G checked in the field (rank-local shard, oracle = checker only)
    verified = None
    if rank == 0 and k <= 26 and world == 1:   # ~3 s of host time at 2^26 (one Horner pass over the scalars), outside the timed region
        from oracle import cref
        sc_host = scalars.cpu().numpy().view(np.uint64)
        p_tau = cref.eval_polynomial(sc_host, tau_m)
        want = cref.g1_to_affine(cref.g1_mul(cref.g1_generator(), p_tau))
        verified = bool((np.asarray(result)[:8] == want).all())

    # N > 1: every rank evaluates its slice of the polynomial in the field, p_r(tau) tau^(r n) (oracle = checker only, outside the timed region), the 32-byte
    # values are gathered, and rank 0 checks  result == (sum_r tau^(r n) p_r(tau)) G = p(tau) G  for the ONE polynomial p the ranks hold together
    if world > 1 and k <= 26:
        from oracle import cref
        e_r = cref.f_mul(cref.FR, cref.eval_polynomial(scalars.cpu().numpy().view(np.uint64), tau_m), h2.fr(pow(tau, rank * n, h2.R_MOD)))
        mine = torch.from_numpy(e_r.view(np.uint8).copy()).to("cpu" if share else dev)
        allv = torch.empty(world * 32, dtype=torch.uint8, device=mine.device)
        dist.all_gather_into_tensor(allv, mine)
        if rank == 0:
            ev = allv.cpu().numpy().view(np.uint64).reshape(world, 4)
            tot = ev[0]
            for r_ in range(1, world):
                tot = cref.f_add(cref.FR, tot, ev[r_])
            want = cref.g1_to_affine(cref.g1_mul(cref.g1_generator(), tot))
            verified = bool((np.asarray(result)[:8] == want).all())

    ms_per_step = dt / args.steps * 1e3
    pairs_per_s = n_total * args.steps / dt
    shared_buckets = pre_ms is not None
    # N*W bucket additions + 2 * 2^(c-1) running-sum additions per bucket set (W sets per shard, or ONE with precomputed window tables)
    devs = args.gpus if single else world
    adds_per_msm = n_total * W + devs * (1 if shared_buckets else W) * (1 << c)
    value = adds_per_msm * args.steps / dt

    # ---- secondary: NTT fwd + inv at 2^k on this rank (replica), device resident
    ntt = None
    if not args.no_ntt:
        dom = h2.EvaluationDomain(2, k)
        poly = rand_scalars(1 << k, 0x5343524F4C4C0003, dev)
        orig = poly[:4096].clone()
        check(lib.mi355_profile_reset()); check(lib.mi355_profile_enable(1))
        dom.coeff_to_lagrange(poly); dom.lagrange_to_coeff(poly)  # warm-up + round trip check
        torch.cuda.synchronize()
        rt_ok = bool(torch.equal(poly[:4096], orig))
        check(lib.mi355_profile_reset())
        torch.cuda.synchronize(); t1 = time.perf_counter()
        reps = max(1, args.steps)
        for _ in range(reps):
            dom.coeff_to_lagrange(poly); dom.lagrange_to_coeff(poly)
        check(lib.mi355_synchronize()); torch.cuda.synchronize()
        dt_ntt = (time.perf_counter() - t1) / (2 * reps)
        check(lib.mi355_profile_enable(0))
        pass_ms, pass_cnt = prof("ntt_pass")
        bf = (1 << k) // 2 * k
        ntt_traffic, ntt_traffic_src = recorded_traffic(f"ntt_k{k}_hbm_bytes_per_transform", "ntt")   # HBM bytes per transform from the separate --pmc passes
        ntt = {"log_n": k, "ms_per_transform": dt_ntt * 1e3, "butterflies_per_s": bf / dt_ntt, "roundtrip_ok": rt_ok,
               "passes_per_transform": pass_cnt / (2 * reps),
               "roofline": {"bound": "hbm", "achieved": 64.0 * (1 << k) / dt_ntt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": 64.0 * (1 << k) / dt_ntt / 1e9 / HBM_PEAK_GBS, "traffic": ntt_traffic, "traffic_source": ntt_traffic_src,
                            "note": "algorithmic bytes = 64*N per transform (SURVEY 8d); whole-transform time, all passes"}}
        # eval_polynomial (step 9 of create_proof): streaming, 1 multiplication per 32-byte coefficient -> the HBM-bound kernel of the path
        pt = h2.fr(0x1234567890ABCDEF)
        h2.eval_polynomial(poly, pt)
        torch.cuda.synchronize(); t5 = time.perf_counter()
        for _ in range(5):
            h2.eval_polynomial(poly, pt)
        dt_ev = (time.perf_counter() - t5) / 5
        ntt["eval_polynomial"] = {"log_n": k, "ms": dt_ev * 1e3, "roofline": {"bound": "hbm", "achieved": 32.0 * (1 << k) / dt_ev / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": 32.0 * (1 << k) / dt_ev / 1e9 / HBM_PEAK_GBS, "note": "algorithmic bytes = 32 B per coefficient, host wall time incl. the 32-byte result copy"}}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            # CPU baseline of the transform (rank 0, N = 1): the oracle's restatement of best_fft (bit-reverse + radix-2 layers, thread split) on
            # the SAME 2^k coefficients, all usable host threads (~10 s at 2^26); its output doubles as the checker of the device transform
            from oracle import cref
            cores = cref.usable_cpus()   # affinity mask and cgroup quota, not the host's hardware threads
            ks_ = min(k, args.cpu_sample_log)
            sample = poly[: 1 << ks_].cpu().numpy().view(np.uint64).reshape(1 << ks_, 4).copy()
            dom_s = h2.EvaluationDomain(2, ks_)
            t8 = time.perf_counter(); want_f = cref.best_fft(sample, dom_s.omega, ks_, threads=cores); dt_c = time.perf_counter() - t8
            ntt["cpu_baseline"] = {"value": (1 << ks_) // 2 * ks_ / dt_c, "unit": "butterflies/s", "cores": cores, "host_hw_threads": os.cpu_count(), "kind": "port",
                                   "sample": f"best_fft restatement (oracle/bn254_oracle.c, pthreads) on 2^{ks_} coefficients of the same data" + (" (the full workload)" if ks_ == k else "") + f", {dt_c:.2f} s wall"}
            if ks_ == k:
                dom.coeff_to_lagrange(poly)
                ntt["full_vector_equals_cpu_baseline"] = bool((poly.cpu().numpy().view(np.uint64).reshape(1 << k, 4) == want_f).all())
            del sample, want_f
        del poly

    extra = {}
    if world == 1 and not args.no_host_api:
        # the entry point the Rust shim calls: scalars in ordinary (pageable) host memory cross PCIe inside the call.  Never `value`.
        sc_host = scalars.cpu().numpy().view(np.uint64)
        check(lib.mi355_msm_g1_host(handle.value, 0, ptr(sc_host), n, ptr(out)))   # warm-up (allocates the staging buffer)
        host_ok = bool((out == result).all())
        check(lib.mi355_profile_reset()); check(lib.mi355_profile_enable(1))
        t3 = time.perf_counter()
        for _ in range(3):
            check(lib.mi355_msm_g1_host(handle.value, 0, ptr(sc_host), n, ptr(out)))
        host_ms = (time.perf_counter() - t3) / 3 * 1e3
        check(lib.mi355_profile_enable(0))
        dv, ex, shd, sl = C.c_int(), C.c_char_p(), C.c_int(), C.c_int()
        check(lib.mi355_msm_last_run(C.byref(dv), C.byref(ex), C.byref(shd), C.byref(sl)))
        extra["host_api"] = {"ms_per_commit_pcie_inclusive": host_ms, "vs_resident": host_ms / ms_per_step, "point_range_slices": sl.value,
                             "same_result_as_resident": host_ok, "scalars": "pageable host memory (numpy), 32 B x 2^%d" % k,
                             "phase_ms_sum_over_slices": {p_: prof(p_)[0] / 3 for p_ in ("msm_digits", "msm_sort", "msm_accumulate", "msm_reduce", "msm_total")}}
        del sc_host
    if world == 1 and pre_ms is not None and not args.no_table_free:
        # the memory-lean configuration: no window tables (W x the basis of HBM saved), per-window bucket sets and the Horner tail
        check(lib.mi355_msm_set_window_bits(-1))
        check(lib.mi355_msm_g1_dev(handle.value, 0, ptr(scalars), n, ptr(out)))
        nt_ok = bool((out == result).all())
        torch.cuda.synchronize(); t7 = time.perf_counter()
        for _ in range(2):
            check(lib.mi355_msm_g1_dev(handle.value, 0, ptr(scalars), n, ptr(out)))
        nt_ms = (time.perf_counter() - t7) / 2 * 1e3
        c2, w2, e2 = C.c_int(), C.c_int(), C.c_uint64()
        check(lib.mi355_msm_last_plan(C.byref(c2), C.byref(w2), C.byref(e2)))
        check(lib.mi355_msm_set_window_bits(0))
        extra["without_window_tables"] = {"ms_per_commit": nt_ms, "window_bits": c2.value, "windows": w2.value, "same_result": nt_ok}
    if proof_mix is not None:
        extra["proof_mix"] = proof_mix
    if world == 1 and k <= 26 and not args.no_witness_like:
        # the second scalar distribution the survey asks for: mostly zeros / tiny values (giant buckets, few entries)
        wl = witness_like_scalars(n, 0x5343524F4C4C0004, dev, h2)
        check(lib.mi355_msm_g1_dev(handle.value, 0, ptr(wl), n, ptr(out)))
        torch.cuda.synchronize(); t6 = time.perf_counter()
        for _ in range(3):
            check(lib.mi355_msm_g1_dev(handle.value, 0, ptr(wl), n, ptr(out)))
        dt_w = (time.perf_counter() - t6) / 3
        check(lib.mi355_profile_reset()); check(lib.mi355_profile_enable(1))
        check(lib.mi355_msm_g1_dev(handle.value, 0, ptr(wl), n, ptr(out)))
        check(lib.mi355_profile_enable(0))
        ph_w = {p_: prof(p_)[0] for p_ in ("msm_digits", "msm_sort", "msm_accumulate", "msm_reduce", "msm_total")}
        cw, ww, ew = C.c_int(), C.c_int(), C.c_uint64()
        check(lib.mi355_msm_last_plan(C.byref(cw), C.byref(ww), C.byref(ew)))
        ok_w = None
        if k <= 26:   # ~3 s of host time at 2^26 (one Horner pass), outside every timed region
            from oracle import cref
            ok_w = bool((np.asarray(out)[:8] == cref.g1_to_affine(cref.g1_mul(cref.g1_generator(), cref.eval_polynomial(wl.cpu().numpy().view(np.uint64), tau_m)))).all())
        extra["witness_like"] = {"ms_per_commit": dt_w * 1e3, "pairs_per_s": n / dt_w, "verified_against_field_check": ok_w, "msm_phase_ms": ph_w,
                                 "distribution": "60% zero, 20% in 1..255, 10% 64-bit, 10% uniform"}
        del wl

    if world == 1 and not args.no_sizes:
        # north_star asks for k = 20 / 24 / 26: the two smaller sizes on PREFIX VIEWS of the same registered basis (what load_params_map's
        # clone + downsize amounts to), each with the tables mi355_srs_precompute picks for its length; same scalars, same checks
        from oracle import cref
        sizes = {}
        for ks in (20, 24):
            if ks >= k:
                continue
            ns = 1 << ks
            hp = C.c_uint64()
            check(lib.mi355_srs_register_prefix(handle.value, ns, C.byref(hp)))
            if pre_ms is not None:
                check(lib.mi355_srs_precompute(hp.value, 0, 0))
            sc_s = scalars[:ns]
            check(lib.mi355_msm_g1_dev(hp.value, 0, ptr(sc_s), ns, ptr(out)))
            reps_s = 20 if ks <= 20 else 5
            torch.cuda.synchronize(); t9 = time.perf_counter()
            for _ in range(reps_s):
                check(lib.mi355_msm_g1_dev(hp.value, 0, ptr(sc_s), ns, ptr(out)))
            dt_s = (time.perf_counter() - t9) / reps_s
            # phase breakdown from a separate short pass (the HIP events of the profiler cost a few microseconds per phase: kept out of the timing)
            check(lib.mi355_profile_reset()); check(lib.mi355_profile_enable(1))
            for _ in range(3):
                check(lib.mi355_msm_g1_dev(hp.value, 0, ptr(sc_s), ns, ptr(out)))
            check(lib.mi355_profile_enable(0))
            ph_s = {p_: prof(p_)[0] / 3 for p_ in ("msm_digits", "msm_sort", "msm_accumulate", "msm_reduce", "msm_total")}
            cs, ws, es = C.c_int(), C.c_int(), C.c_uint64()
            check(lib.mi355_msm_last_plan(C.byref(cs), C.byref(ws), C.byref(es)))
            dv, ex, shd, sl = C.c_int(), C.c_char_p(), C.c_int(), C.c_int()
            check(lib.mi355_msm_last_run(C.byref(dv), C.byref(ex), C.byref(shd), C.byref(sl)))
            ok_s = bool((np.asarray(out)[:8] == cref.g1_to_affine(cref.g1_mul(cref.g1_generator(), cref.eval_polynomial(sc_s.cpu().numpy().view(np.uint64), tau_m)))).all())
            adds_s = ns * ws.value + (1 if shd.value else ws.value) * (1 << cs.value)
            rec = {"msm_ms_per_commit": dt_s * 1e3, "pairs_per_s": ns / dt_s, "g1_adds_per_s": adds_s / dt_s, "window_bits": cs.value, "windows": ws.value,
                   "srs_window_tables": bool(shd.value), "verified_against_field_check": ok_s, "msm_phase_ms": ph_s,
                   "msm_roofline_frac_hbm": 96.0 * ns / dt_s / 1e9 / HBM_PEAK_GBS}
            check(lib.mi355_srs_release(hp.value))
            if not args.no_cpu_baseline:
                # the reference CPU path beside this size too (north_star: "k = 20 / 24 / 26 ... next to the reference CPU path ... in the same run"): same pairs
                cores_s = cref.usable_cpus()
                g_s = g[: ns * 64].cpu().numpy().view(np.uint64).reshape(ns, 8); sc_h = sc_s.cpu().numpy().view(np.uint64)
                tc = time.perf_counter(); ref_s = cref.best_multiexp(sc_h, g_s, threads=cores_s); dt_c = time.perf_counter() - tc
                chunk_s = max(1, ns // cores_s); c_s = int(np.ceil(np.log(chunk_s))) if chunk_s >= 32 else 3; seg_s = 256 // c_s + 1
                rec["cpu_baseline"] = {"msm": {"value": (ns * seg_s + (ns // chunk_s) * seg_s * 2 * ((1 << c_s) - 1)) / dt_c, "unit": "G1-adds/s", "pairs_per_s": ns / dt_c, "cores": cores_s, "kind": "port",
                                               "sample": f"best_multiexp restatement on the same 2^{ks} pairs, {dt_c:.2f} s wall", "equals_gpu_result": bool((np.asarray(out)[:8] == cref.g1_to_affine(ref_s)).all())}}
                del g_s, sc_h
            if not args.no_ntt:
                dom_s = h2.EvaluationDomain(2, ks)
                poly_s = rand_scalars(ns, 0x5343524F4C4C0003, dev)
                dom_s.coeff_to_lagrange(poly_s); dom_s.lagrange_to_coeff(poly_s)
                torch.cuda.synchronize(); t10 = time.perf_counter()
                for _ in range(reps_s):
                    dom_s.coeff_to_lagrange(poly_s); dom_s.lagrange_to_coeff(poly_s)
                check(lib.mi355_synchronize()); torch.cuda.synchronize()
                dt_n = (time.perf_counter() - t10) / (2 * reps_s)
                rec.update({"ntt_ms_per_transform": dt_n * 1e3, "butterflies_per_s": ns // 2 * ks / dt_n, "ntt_roofline_frac_hbm": 64.0 * ns / dt_n / 1e9 / HBM_PEAK_GBS})
                if not args.no_cpu_baseline:
                    src_h = poly_s.cpu().numpy().view(np.uint64).reshape(ns, 4).copy()
                    tn = time.perf_counter(); want_s = cref.best_fft(src_h, dom_s.omega, ks, threads=cref.usable_cpus()); dt_cn = time.perf_counter() - tn
                    dom_s.coeff_to_lagrange(poly_s)
                    rec.setdefault("cpu_baseline", {})["ntt"] = {"value": ns // 2 * ks / dt_cn, "unit": "butterflies/s", "cores": cref.usable_cpus(), "kind": "port", "sample": f"best_fft restatement on the same 2^{ks} coefficients, {dt_cn:.2f} s wall",
                                                                "equals_gpu_result": bool((poly_s.cpu().numpy().view(np.uint64).reshape(ns, 4) == want_s).all())}
                    del src_h, want_s
                del poly_s
            sizes["k%d" % ks] = rec
        extra["sizes"] = sizes

    if world == 1 and not args.no_batch_legs and k >= 22:
        # the k = 20 / 21 regime of the real proof: layer 3 commits ~110 columns of 2^21 [REF integration/configs/layer3.config:3-8], layer 0 O(10^3)
        # columns of 2^20 [REF integration/src/capacity_checker.rs:90-92].  One mi355_msm_g1_batch_dev call = the per-column commit loop of
        # create_proof as ONE pass (one reduction tail per batch instead of per column); first and last result checked in the field.
        from oracle import cref
        legs = {}
        for ks, M in ((21, 110), (20, 256)):
            ns = 1 << ks
            hp = C.c_uint64()
            check(lib.mi355_srs_register_prefix(handle.value, ns, C.byref(hp)))
            if pre_ms is not None:
                check(lib.mi355_srs_precompute(hp.value, 0, 0))
            cols = rand_scalars(M * ns, 0x5343524F4C4C0010 + ks, dev)
            arr = (C.c_void_p * M)(*[cols[m * ns:(m + 1) * ns].data_ptr() for m in range(M)])
            outs = np.zeros((M, 12), dtype=np.uint64)
            check(lib.mi355_msm_g1_batch_dev(hp.value, 0, arr, M, ns, ptr(outs)))
            torch.cuda.synchronize(); tb = time.perf_counter()
            for _ in range(2):
                check(lib.mi355_msm_g1_batch_dev(hp.value, 0, arr, M, ns, ptr(outs)))
            dt_b = (time.perf_counter() - tb) / 2
            cb, wb, eb = C.c_int(), C.c_int(), C.c_uint64()
            check(lib.mi355_msm_last_plan(C.byref(cb), C.byref(wb), C.byref(eb)))
            one = np.zeros(12, dtype=np.uint64)
            check(lib.mi355_msm_g1_dev(hp.value, 0, ptr(cols[:ns]), ns, ptr(one)))
            torch.cuda.synchronize(); t1c = time.perf_counter()
            for _ in range(10):
                check(lib.mi355_msm_g1_dev(hp.value, 0, ptr(cols[:ns]), ns, ptr(one)))
            dt_1 = (time.perf_counter() - t1c) / 10
            ok_b = all(bool((outs[m][:8] == cref.g1_to_affine(cref.g1_mul(cref.g1_generator(), cref.eval_polynomial(cols[m * ns:(m + 1) * ns].cpu().numpy().view(np.uint64), tau_m)))).all()) for m in (0, M - 1))
            ok_b = ok_b and bool((outs[0] == one).all())
            legs[f"{M}x2^{ks}"] = {"columns": M, "log_n": ks, "ms_per_batch": dt_b * 1e3, "ms_per_commit": dt_b * 1e3 / M, "pairs_per_s": M * ns / dt_b,
                                   "g1_adds_per_s": eb.value / dt_b, "window_bits": cb.value, "windows": wb.value,
                                   "single_commit_ms": dt_1 * 1e3, "single_commit_pairs_per_s": ns / dt_1, "verified_against_field_check": ok_b}
            check(lib.mi355_srs_release(hp.value))
            del cols
            torch.cuda.empty_cache()
        extra["batched_commitments"] = legs

    # ---- CPU baseline (rank 0, N = 1 only): the restated reference algorithm on a bounded sample of the same workload
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cref
        cores = cref.usable_cpus()   # affinity mask and cgroup quota, not the host's hardware threads
        ks = min(k, args.cpu_sample_log)   # the full 2^26 workload: ~25 s wall on 16 usable CPUs
        ns = 1 << ks
        sc_host = scalars[:ns].cpu().numpy().view(np.uint64)
        g_host = g[: ns * 64].cpu().numpy().view(np.uint64).reshape(ns, 8)
        t2 = time.perf_counter()
        ref = cref.best_multiexp(sc_host, g_host, threads=cores)
        dt_cpu = time.perf_counter() - t2
        chunk = max(1, ns // cores)
        c_cpu = int(np.ceil(np.log(chunk))) if chunk >= 32 else 3
        seg = 256 // c_cpu + 1
        adds_cpu = ns * seg + (ns // chunk) * seg * 2 * ((1 << c_cpu) - 1)
        cpu = {"value": adds_cpu / dt_cpu, "unit": "G1-adds/s", "cores": cores, "host_hw_threads": os.cpu_count(), "kind": "port",
               "sample": f"best_multiexp restatement (oracle/bn254_oracle.c, pthreads, c=ceil(ln(n/threads))={c_cpu}, {seg} segments) on the first 2^{ks} pairs of the same workload, {dt_cpu:.2f} s wall",
               "pairs_per_s": ns / dt_cpu}
        if ks == k:
            verified = bool((np.asarray(result)[:8] == cref.g1_to_affine(ref)).all()) and (verified is not False)

    if rank == 0:
        acc_avg_ms = acc_ms / max(1, acc_cnt)
        pairs_per_launch = n / chunks / (args.gpus if single else 1)   # one accumulate launch processes one chunk of this device's point range, all windows
        achieved = 96.0 * pairs_per_launch / (acc_avg_ms * 1e-3) / 1e9 if acc_avg_ms > 0 else None
        # HBM bytes of one k_msm_accumulate launch from the PMC counters: RECORDED from a separate rocprofv3 --pmc pass over this same
        # command (profiles/r02b_pmc_msm_k26.md), not measured inside this run (counter collection cannot share a run with the timing)
        traffic, traffic_src = (recorded_traffic(f"msm_accumulate_k{k}_hbm_bytes_per_launch", "msm") if (not single and world == 1) else (None, None))
        line = {
            "metric": "BN254 MSM G1-adds/sec at k=%d" % k, "value": value, "unit": "G1-adds/s", "n_gpus": args.gpus if single else world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32 (9x29-bit limbs of 254-bit Montgomery integers; v_mad_u64_u32)", "data": "synthetic",
            "config": {"workload": f"BN254 G1 Pippenger MSM, 2^{k} uniform random scalars x synthetic SRS points, inputs resident in HBM; "
                                   f"point-range shards over {world} GPU(s), RCCL all-gather of 96-B partials",
                       "log_n": k, "window_bits": c, "windows": W, "parallelism": (f"point-range x{args.gpus} inside one process (mi355_init_multi: per-device shards, ncclAllGather of the partials in the library)" if single else f"point-range x{world}, one rank per GPU"),
                       "srs_window_tables": shared_buckets, "srs_precompute_ms_once": pre_ms},
            "pairs_per_s": pairs_per_s, "g1_adds_per_msm": adds_per_msm, "verified_against_field_check": verified,
            "msm_phase_ms": phases, "msm_pipeline_chunks": chunks,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "k_msm_accumulate", "avg_launch_ms": acc_avg_ms,
                         "pairs_per_launch": pairs_per_launch,
                         "alu": {"achieved": (pairs_per_launch * W / (acc_avg_ms * 1e-3)) if acc_avg_ms > 0 else None, "peak": MADD_CHAIN_PEAK, "unit": "G1 mixed additions/s",
                                 "frac": (pairs_per_launch * W / (acc_avg_ms * 1e-3) / MADD_CHAIN_PEAK) if acc_avg_ms > 0 else None,
                                 "source": "profiles/r02_microbench.log (xyzz29 madd chain, 3 waves/SIMD, boost clock); SQ counters: VALU 98 % busy at the sustained 1.93 GHz, 2 193 VALU instructions per addition, profiles/r03_sq_counters.md",
                                 # SURVEY 8d's own definition, next to the chain ceiling: multiplier instructions executed / time / the
                                 # machine's v_mad_u64_u32 issue rate (one per lane per 4 cycles: CUs x 4 SIMDs x 16 lanes x clock, nominal 2.4 GHz)
                                 "mad_rate": {"mads_per_addition": MADS_PER_MADD, "achieved": (MADS_PER_MADD * pairs_per_launch * W / (acc_avg_ms * 1e-3)) if acc_avg_ms > 0 else None,
                                              "peak": torch.cuda.get_device_properties(dev).multi_processor_count * 4 * 16 * 2.4e9, "unit": "v_mad_[ui]64 per second",
                                              "frac": (MADS_PER_MADD * pairs_per_launch * W / (acc_avg_ms * 1e-3) / (torch.cuda.get_device_properties(dev).multi_processor_count * 4 * 16 * 2.4e9)) if acc_avg_ms > 0 else None,
                                              "source": "profiles/r04_accumulate_isa_blocks.md (1 467 multiplier instructions on the common path of one mixed addition); the sustained clock under this kernel is 1.93 GHz, at which the fraction is 0.65"}},
                         "note": "algorithmic bytes = 96 B per (scalar, point) pair x pairs per launch (SURVEY 8d); the kernel is VALU-integer bound, see DESIGN.md"},
            "cpu_baseline": cpu, "ntt": ntt,
        }
        if multi is not None:
            line["multi_gpu"] = multi
        if single and args.gpus > 1:
            dv, ex, shd, sl = C.c_int(), C.c_char_p(), C.c_int(), C.c_int()
            check(lib.mi355_msm_last_run(C.byref(dv), C.byref(ex), C.byref(shd), C.byref(sl)))
            line["multi_gpu"] = {"devices": dv.value, "exchange": (ex.value or b"").decode(), "mode": "one process, mi355_init_multi"}
        line.update(extra)
        # the other two legs of BASELINE.json's metric where the driver's record keeps them: the NTT at 2^k and the three proof proxies,
        # plus one flag that every check of this run passed (MSM field check, NTT round trip / CPU equality, every proof verified from its bytes)
        cfg = line["config"]
        checks = [verified is not False]
        if ntt is not None:
            cfg["ntt_k%d_ms" % k] = round(ntt["ms_per_transform"], 3); cfg["ntt_k%d_butterflies_per_s" % k] = ntt["butterflies_per_s"]; cfg["ntt_roofline_frac_hbm"] = round(ntt["roofline"]["frac"], 4)
            checks += [ntt["roundtrip_ok"], ntt.get("full_vector_equals_cpu_baseline", True)]
        if proof_mix is not None and "chunk_proof_proxy" in proof_mix:
            sep = [proof_mix[x]["resident_ms"] for x in ("chunk_proof_proxy", "batch_proof_proxy", "bundle_proof_proxy")]
            cfg["layer_ms"] = {str(x): (proof_mix[p_][f"layer{x}"].get("resident_ms")) for p_, xs in (("chunk_proof_proxy", (0, 1, 2)), ("batch_proof_proxy", (3, 4)), ("bundle_proof_proxy", (5, 6))) for x in xs}
            checks.append(bool(proof_mix["every_proof_verified"]))
            # the chunk proxy in the REFERENCE's process shape: ChunkProver is ONE process holding layers 0 + 1 + 2 [REF integration/src/prove.rs:30-43]; the sum of
            # three single-layer processes (each with the whole HBM to itself) is the flattering figure and is reported beside it, never as it
            cpp_ = proof_mix.get("chunk_prover_process")
            one = cpp_.get("round_ms") if (cpp_ is not None and cpp_.get("ok")) else None
            cfg["proxies_ms"] = [one if one is not None else sep[0], sep[1], sep[2]]
            cfg["proxies_separate_processes_ms"] = sep
            cfg["proxies_shape"] = ("[chunk = L0+L1+L2 in ONE prover process under plan_residency, " if one is not None else "[chunk = L0+L1+L2 as three single-layer processes (one-process run skipped or failed), ") + \
                "batch = L3+L4, bundle = L5+L6 as single-layer processes]; create_proof per layer from its PlonkProtocol (layers 2 / 4: the reference's own), every proof verified from its bytes"
            if cpp_ is not None:
                cfg["chunk_prover_process_ms"] = cpp_.get("round_ms"); cfg["chunk_prover_process_peak_hbm_gib"] = (cpp_.get("hbm") or {}).get("peak_used_gib")
                checks.append(bool(cpp_.get("ok")))
        if proof_mix is not None and "chunk_proof_proxy" in proof_mix:
            rp = released_proofs_check()     # the verifier that accepted this run's proofs, on the reference's own released proofs (about 10 s of CPU, after all timing)
            if rp is not None:
                cfg["verifier_accepts_released_reference_proofs"] = "%d of %d" % rp
                checks.append(rp[0] == rp[1])
        cfg["all_checks"] = all(bool(c) for c in checks)
        emit(line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
