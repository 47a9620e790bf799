#!/usr/bin/env python3
"""gen_glv_constants.py -- derives the GLV constants of BN254 G1 used by scroll-prover_amd/csrc/glv.hpp and prints them as C++ initialisers.

  lambda : primitive cube root of unity in Fr,  beta : primitive cube root of unity in Fq,  with  lambda * (x, y) = (beta x, y)  on G1
  (a1, b1), (a2, b2) : short basis of the lattice {(a, b) : a + b lambda = 0 mod r}  (extended Euclid on (r, lambda), Gallant-Lambert-Vanstone)
  g1 = floor(2^256 b2 / r), g2 = floor(2^256 (-b1) / r) : c_i = (k g_i) >> 256 approximate round(b2 k / r), round(-b1 k / r)
  k = k1 + lambda k2 (mod r) with k1 = k - c1 a1 - c2 a2, k2 = -c1 b1 - c2 b2, |k1|, |k2| < 2^128 (checked over random and extreme k below).
Everything is re-derived from r and p; tests/test_host_logic.py re-runs this derivation and compares it with the header."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyref

R, P = pyref.R_MOD, pyref.P_MOD


def cube_roots(m, gen):
    z = pow(gen, (m - 1) // 3, m)
    assert z != 1 and pow(z, 3, m) == 1
    return z, z * z % m


def derive():
    G = pyref.G1_GEN
    lams = cube_roots(R, 7)
    betas = cube_roots(P, 3) if pow(3, (P - 1) // 3, P) != 1 else cube_roots(P, 5)
    pair = None
    for lam in lams:
        Q = pyref.g1_mul(G, lam)
        for beta in betas:
            if Q == (beta * G[0] % P, G[1]):
                pair = (lam, beta)
    assert pair, "no (lambda, beta) pair found"
    lam, beta = pair
    # extended Euclid on (r, lambda): r_i = s_i r + t_i lambda
    rs, ts = [R, lam], [0, 1]
    while rs[-1] != 0:
        q = rs[-2] // rs[-1]
        rs.append(rs[-2] - q * rs[-1]); ts.append(ts[-2] - q * ts[-1])
    l = max(i for i in range(len(rs)) if rs[i] * rs[i] >= R)       # r_l >= sqrt(r) > r_(l+1)
    a1, b1 = rs[l + 1], -ts[l + 1]
    cand = [(rs[l], -ts[l]), (rs[l + 2], -ts[l + 2])]
    a2, b2 = min(cand, key=lambda v: v[0] * v[0] + v[1] * v[1])
    for a, b in ((a1, b1), (a2, b2)):
        assert (a + b * lam) % R == 0
    det = a1 * b2 - a2 * b1
    assert abs(det) == R
    if det < 0:   # orient the basis so that det = +r (c1 = b2 k / r, c2 = -b1 k / r)
        a2, b2 = -a2, -b2
    g1 = (b2 << 256) // R if b2 >= 0 else -((-b2 << 256) // R)
    g2 = (-b1 << 256) // R if -b1 >= 0 else -((b1 << 256) // R)
    return dict(lam=lam, beta=beta, a1=a1, b1=b1, a2=a2, b2=b2, g1=g1, g2=g2)


def decompose(k, c):
    """the integer formulas of glv_decompose (glv.hpp): magnitudes of g1 / g2 with their signs applied afterwards"""
    c1 = (k * abs(c["g1"])) >> 256
    c2 = (k * abs(c["g2"])) >> 256
    if c["g1"] < 0: c1 = -c1
    if c["g2"] < 0: c2 = -c2
    k1 = k - c1 * c["a1"] - c2 * c["a2"]
    k2 = -c1 * c["b1"] - c2 * c["b2"]
    return k1, k2


def limbs32(x, n):
    return ", ".join("0x%08xu" % ((x >> (32 * i)) & 0xffffffff) for i in range(n))


if __name__ == "__main__":
    c = derive()
    rng = random.Random(5)
    worst = 0
    for k in [0, 1, 2, R - 1, R - 2, R // 2, R // 3, (1 << 253), c["lam"], R - c["lam"]] + [rng.randrange(R) for _ in range(200000)]:
        k1, k2 = decompose(k, c)
        assert (k1 + c["lam"] * k2 - k) % R == 0
        worst = max(worst, abs(k1), abs(k2))
    print("// worst |k_i| over the sample: 2^%.3f" % (worst.bit_length() - 1 + (worst / (1 << (worst.bit_length() - 1)) - 1)))
    for name in ("lam", "beta"):
        print("// %s = 0x%x" % (name, c[name]))
    for name in ("a1", "b1", "a2", "b2", "g1", "g2"):
        v = c[name]
        print("// %s = %s0x%x" % (name, "-" if v < 0 else "", abs(v)))
    print("beta_mont29 (beta * 2^261 mod p, 9 x 29-bit limbs): {%s}" % ", ".join("0x%xu" % ((c["beta"] * (1 << 261) % P >> (29 * i)) & ((1 << 29) - 1)) for i in range(9)))
    print("beta_mont (beta * 2^256 mod p, 8 x 32): {%s}" % limbs32(c["beta"] * (1 << 256) % P, 8))
    for name in ("a1", "b1", "a2", "b2"):
        print("%s: neg=%d mag={%s}" % (name, c[name] < 0, limbs32(abs(c[name]), 4)))
    for name in ("g1", "g2"):
        print("%s: neg=%d mag={%s}" % (name, c[name] < 0, limbs32(abs(c[name]), 8)))
