#!/usr/bin/env python3
"""fuzz_plonk.py [iterations] [seed] -- random circuit SHAPES through the protocol-driven prover: the synthetic-inner generator with random column / lookup / permutation
counts, degrees 5 and 9, lookup widths 1-3, and the halo2-base rule on random configs, at k = 6 ... 10, with random replay options (recomputed cosets, device slots, sparse
uploads, prefix groups on / off / forced).  Every case: the device's proof bytes must equal the CPU restatement's (oracle/plonk.py) and the verifier must accept them.
Exercises the plan compiler's cost model, temporaries, relocation and launch splitting on trees no fixed test holds.  Prints one line per case and a summary."""
import json, os, random, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
zk = ge.load_package()
from oracle import plonk
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
P = zk.protocols
bad = 0
t_all = time.time()
for it in range(iters):
    k = rng.randint(6, 10)
    d = tempfile.mkdtemp(prefix="fuzz_plonk_")
    path = os.path.join(d, "p.json")
    if rng.random() < 0.6:
        degree = rng.choice((5, 9)); W = rng.randint(1, 3); lookups = rng.randint(0, 12); groups = min(lookups, 8)
        n_free = 4 + W * groups
        advice = n_free + 2 + rng.randint(0, 40); fixed = W + 2 + rng.randint(0, 6); perm = rng.randint(degree - 2, min(advice + 1, 40))
        shape = dict(advice=advice, fixed=fixed, lookups=lookups, perm_columns=perm, degree=degree, lookup_width=W, num_instance=rng.randint(1, 5))
        if lookups == 0:
            continue   # the protocol reader expects the three witness phases of the fixtures (a lookup-free circuit has no phase 1)
        proto = P.synthetic_inner_protocol(k=k, **shape); proto["layer"] = 0
        desc = f"synthetic k={k} {shape}"
    else:
        A = rng.choice((1, 1, 2, 3, 5, 9)); LA = rng.randint(1, min(3, A)) if A > 1 else 1
        cfg = {"degree": k, "num_advice": [A], "num_lookup_advice": [LA], "num_fixed": rng.randint(1, 2), "lookup_bits": rng.randint(2, k - 1)}
        proto = P.halo2_base_protocol(cfg, rng.randint(1, 12)); proto["layer"] = rng.randint(1, 6)
        desc = f"halo2-base k={k} {cfg}"
    json.dump(proto, open(path, "w"))
    args = ["--dump-inputs", "--proofs", str(rng.choice((1, 2))), "--seed", str(rng.randint(1, 1 << 30)), "--fill", str(rng.choice((0.3, 0.9, 1.0)))]
    env = {"MI355_PLAN_PREFIX_MIN": rng.choice(("16", "0", "2", "4"))}
    if rng.random() < 0.3: args += ["--pk-cosets", "on-the-fly"]
    if rng.random() < 0.3: args += ["--assign-density", str(rng.choice((0.2, 0.6)))]
    if rng.random() < 0.3 and k >= 9: args += ["--sparse-uploads"]
    if rng.random() < 0.3: args += ["--no-packed-multiplicities"]
    if rng.random() < 0.3: args += ["--upload-threads", "3", "--early-intt", "1"]
    if rng.random() < 0.25:
        args += ["--devices", str(rng.choice((2, 3, 8)))]; env.update({"MI355_ALLOW_DUP_DEVICES": "1", "MI355_SHARD_MIN_LOG": "5"})
    rec = zk.replay.run(proto["layer"], out_dir=d, args=args, env=env, protocol_file=path, timeout=600)
    ok = False; why = ""
    if not rec.get("ok"):
        why = "replay failed: " + str(rec.get("error"))[-300:]
    else:
        try:
            inp, man = plonk.ProofInputs.load(d)
            vk = plonk.keygen_vk(inp.pr, inp.pre, inp.tau)
            want = plonk.prove(inp, vk, transcript=rec["transcript"])
            same = rec["proof"] == want and rec["vk"] == vk
            ver = plonk.verify(inp.pr, rec["vk"], inp.instances, rec["proof"], inp.tau, transcript=rec["transcript"])["ok"]
            ok = same and ver
            if not ok:
                first = next((i // 32 for i in range(0, min(len(want), len(rec["proof"])), 32) if want[i:i + 32] != rec["proof"][i:i + 32]), None)
                why = f"bytes_equal={same} (first differing word {first}) verified={ver}"
        except AssertionError as e:
            why = "restatement refused the instance: " + str(e)
    bad += 0 if ok else 1
    print(("ok  " if ok else "FAIL"), desc, " ".join(args[2:]), env.get("MI355_PLAN_PREFIX_MIN"), (rec.get("plan") or ""), why, flush=True)
    import shutil; shutil.rmtree(d, ignore_errors=True)
print(f"fuzz_plonk: {iters} iterations, {bad} failures, {time.time() - t_all:.0f} s")
sys.exit(1 if bad else 0)
