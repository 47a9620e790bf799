#!/usr/bin/env python3
"""
make_regression.py -- regenerates tests/golden/regression.json: end-to-end REGRESSION vectors at the MSM / NTT boundary.

These are NOT reference vectors (the reference holds no MSM/NTT known answers, SURVEY 8c: parity stays "unpinned" at that
boundary); they freeze what the two independent oracles (pure-Python big integers `oracle/pyref.py`, computed here, and the C
restatement) agree on today, so that a later change to an oracle or to a kernel that silently alters results is caught against
committed bytes.  Inputs come from splitmix64 with the seeds of SURVEY 8d; everything is small enough for big-int Python.

    python tests/golden/make_regression.py        (CPU only, ~10 s)
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyref  # noqa: E402

R = pyref.R_MOD
MASK = (1 << 64) - 1


def splitmix64(seed):
    x = seed & MASK
    while True:
        x = (x + 0x9E3779B97F4A7C15) & MASK
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
        yield z ^ (z >> 31)


def scalars(seed, n):
    g = splitmix64(seed)
    return [(next(g) | (next(g) << 64) | (next(g) << 128) | (next(g) << 192)) % R for _ in range(n)]


out = {"_generated_by": "tests/golden/make_regression.py", "_note": "regression vectors frozen from oracle/pyref.py; not reference KATs"}
k, n = 5, 32
tau = scalars(0x5343524F4C4C0001, 1)[0]
w = pyref.omega(k)
g = [pyref.g1_mul(pyref.G1_GEN, pow(tau, i, R)) for i in range(n)]
gl = [pyref.g1_mul(pyref.G1_GEN, s) for s in pyref.lagrange_scalars(k, tau)]
sc = scalars(0x5343524F4C4C0002, n)
sc[3] = 0; sc[4] = 1; sc[5] = R - 1
evals = sc
coeffs = pyref.intt(evals, w)
commit = pyref.msm(coeffs, g)
commit_l = pyref.msm(evals, gl)
assert commit == commit_l == pyref.g1_mul(pyref.G1_GEN, pyref.eval_poly(coeffs, tau))
out["srs"] = {"k": k, "tau_seed": "0x5343524F4C4C0001", "g_compressed": [pyref.g1_compress(P).hex() for P in g],
              "g_lagrange_compressed": [pyref.g1_compress(P).hex() for P in gl]}
out["msm"] = {"scalar_seed": "0x5343524F4C4C0002", "scalars_canonical_hex": [format(s, "064x") for s in sc],
              "commit_lagrange_compressed": pyref.g1_compress(commit_l).hex(),
              "coeffs_sha256": hashlib.sha256(b"".join(c.to_bytes(32, "little") for c in coeffs)).hexdigest()}
k2 = 8
a = scalars(0x5343524F4C4C0003, 1 << k2)
f = pyref.ntt(a, pyref.omega(k2))
out["ntt"] = {"k": k2, "seed": "0x5343524F4C4C0003", "input_sha256": hashlib.sha256(b"".join(v.to_bytes(32, "little") for v in a)).hexdigest(),
              "output_sha256_canonical_le": hashlib.sha256(b"".join(v.to_bytes(32, "little") for v in f)).hexdigest(),
              "output_first4_hex": [format(v, "064x") for v in f[:4]]}
ext = pyref.coeff_to_extended(a[:16], 4, 6)
out["coset"] = {"k": 4, "extended_k": 6, "output_sha256_canonical_le": hashlib.sha256(b"".join(v.to_bytes(32, "little") for v in ext)).hexdigest()}
json.dump(out, open(os.path.join(HERE, "regression.json"), "w"), indent=1)
print("wrote regression.json:", out["msm"]["commit_lagrange_compressed"], out["ntt"]["output_sha256_canonical_le"][:16])
