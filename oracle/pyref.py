"""
pyref.py -- second, independent CPU oracle in pure-Python big integers.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).
Shares no code with oracle/bn254_oracle.c: canonical (non-Montgomery) integers, affine
formulas with pow(x, -1, p), textbook DFT.  Two oracles that agree, plus the golden
vectors decoded from the reference's fixtures (tests/golden/), stand in for the
reference's Rust arithmetic, which is not in /root/reference and cannot be built here
(halo2_proofs@e5ddf67 / halo2curves@112f5b9, [REF Cargo.lock:1886-1888,1911-1913]).

Conventions restated from SURVEY.md §8a-0: Fr/Fq serialise as 4 x u64 LE Montgomery limbs
(R = 2^256); G1Affine identity = (0, 0).
"""
from __future__ import annotations

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # Fr
P_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # Fq
MONT_R = 1 << 256
FR_S = 28
FR_GENERATOR = 7
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (R_MOD - 1) >> FR_S, R_MOD)
# halo2curves bn256::Fr::ZETA [EXT-recalled src/bn256/fr.rs]; checked to be a primitive cube root in tests
FR_ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23
G1_B = 3
G1_GEN = (1, 2)
MASK64 = (1 << 64) - 1


# ----------------------------------------------------------------- encodings
def to_limbs(x: int):
    return [(x >> (64 * i)) & MASK64 for i in range(4)]


def from_limbs(l) -> int:
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def to_mont(x: int, m: int) -> int:
    return (x * MONT_R) % m


def from_mont(x: int, m: int) -> int:
    return (x * pow(MONT_R, -1, m)) % m


def mont_limbs(x: int, m: int):
    return to_limbs(to_mont(x % m, m))


def omega(k: int) -> int:
    """omega_k = ROOT_OF_UNITY^(2^(S-k)); EvaluationDomain::new [EXT-recalled poly/domain.rs]."""
    assert 0 <= k <= FR_S
    return pow(FR_ROOT_OF_UNITY, 1 << (FR_S - k), R_MOD)


# ----------------------------------------------------------------- G1 (affine, canonical ints; None = identity)
def g1_is_on_curve(P) -> bool:
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - G1_B) % P_MOD == 0


def g1_neg(P):
    return None if P is None else (P[0], (-P[1]) % P_MOD)


def g1_add(P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % P_MOD == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, P_MOD) % P_MOD
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P_MOD) % P_MOD
    x3 = (lam * lam - x1 - x2) % P_MOD
    y3 = (lam * (x1 - x3) - y1) % P_MOD
    return (x3, y3)


def g1_mul(P, k: int):
    k %= R_MOD
    acc = None
    add = P
    while k:
        if k & 1:
            acc = g1_add(acc, add)
        add = g1_add(add, add)
        k >>= 1
    return acc


def msm(scalars, points):
    acc = None
    for s, P in zip(scalars, points):
        acc = g1_add(acc, g1_mul(P, s))
    return acc


def g1_affine_to_limbs(P):
    """-> (x limbs, y limbs) Montgomery, as G1Affine sits in memory."""
    if P is None:
        return [0] * 4, [0] * 4
    return mont_limbs(P[0], P_MOD), mont_limbs(P[1], P_MOD)


def g1_affine_from_limbs(xl, yl):
    x, y = from_mont(from_limbs(xl), P_MOD), from_mont(from_limbs(yl), P_MOD)
    return None if (x == 0 and y == 0) else (x, y)


def g1_jacobian_from_limbs(xl, yl, zl):
    X, Y, Z = (from_mont(from_limbs(v), P_MOD) for v in (xl, yl, zl))
    if Z == 0:
        return None
    zi = pow(Z, -1, P_MOD)
    return (X * zi * zi % P_MOD, Y * zi * zi * zi % P_MOD)


def g1_compress(P) -> bytes:
    if P is None:
        return bytes(32)
    v = P[0] | ((P[1] & 1) << 254)
    return v.to_bytes(32, "little")


def g1_decompress(b: bytes):
    v = int.from_bytes(b, "little")
    sign = (v >> 254) & 1
    x = v & ((1 << 254) - 1)
    if x == 0 and sign == 0:
        return None
    assert x < P_MOD
    y2 = (x * x * x + G1_B) % P_MOD
    y = pow(y2, (P_MOD + 1) // 4, P_MOD)
    assert y * y % P_MOD == y2, "not on curve"
    if (y & 1) != sign:
        y = P_MOD - y
    return (x, y)


# ----------------------------------------------------------------- NTT
def dft(a, w: int):
    n = len(a)
    return [sum(a[j] * pow(w, i * j, R_MOD) for j in range(n)) % R_MOD for i in range(n)]


def ntt(a, w: int):
    """recursive radix-2, natural in -> natural out (the contract of best_fft)."""
    n = len(a)
    if n == 1:
        return list(a)
    ev = ntt(a[0::2], w * w % R_MOD)
    od = ntt(a[1::2], w * w % R_MOD)
    out = [0] * n
    t = 1
    for i in range(n // 2):
        x = od[i] * t % R_MOD
        out[i] = (ev[i] + x) % R_MOD
        out[i + n // 2] = (ev[i] - x) % R_MOD
        t = t * w % R_MOD
    return out


def intt(a, w: int):
    n = len(a)
    ninv = pow(n, -1, R_MOD)
    return [x * ninv % R_MOD for x in ntt(a, pow(w, -1, R_MOD))]


def coeff_to_extended(coeffs, k: int, ext_k: int, zeta: int = FR_ZETA):
    """EvaluationDomain::coeff_to_extended [EXT-recalled]: pad, a[i] *= zeta^(i%3), fft with extended omega."""
    n, en = 1 << k, 1 << ext_k
    a = list(coeffs) + [0] * (en - n)
    z = [1, zeta, zeta * zeta % R_MOD]
    a = [x * z[i % 3] % R_MOD for i, x in enumerate(a)]
    return ntt(a, omega(ext_k))


def eval_poly(coeffs, x: int) -> int:
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R_MOD
    return acc


def lagrange_scalars(k: int, tau: int):
    """L_i(tau) = omega^i (tau^n - 1) / (n (tau - omega^i))  (ParamsKZG::setup [EXT-recalled])."""
    n = 1 << k
    w = omega(k)
    tn1 = (pow(tau, n, R_MOD) - 1) % R_MOD
    ninv = pow(n, -1, R_MOD)
    out = []
    wi = 1
    for _ in range(n):
        out.append(wi * tn1 % R_MOD * ninv % R_MOD * pow((tau - wi) % R_MOD, -1, R_MOD) % R_MOD)
        wi = wi * w % R_MOD
    return out


# ----------------------------------------------------------------- G2 (affine over Fq2 = Fq[u]/(u^2+1), canonical ints; None = identity)
# Second, independent statement of the twist arithmetic (affine chord-and-tangent with explicit Fq2 inversions; the C oracle uses
# Jacobian coordinates).  G2 appears on this path only as g2 / s_g2 = tau * G2 of ParamsKZG.
def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P_MOD, (a[0] * b[1] + a[1] * b[0]) % P_MOD)


def f2_add(a, b):
    return ((a[0] + b[0]) % P_MOD, (a[1] + b[1]) % P_MOD)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P_MOD, (a[1] - b[1]) % P_MOD)


def f2_inv(a):
    n = pow((a[0] * a[0] + a[1] * a[1]) % P_MOD, -1, P_MOD)
    return (a[0] * n % P_MOD, (-a[1]) * n % P_MOD)


G2_B = f2_mul((3, 0), f2_inv((9, 1)))          # 3 / (9 + u)
G2_GEN = ((0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED, 0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2),
          (0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA, 0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B))


def g2_is_on_curve(Q) -> bool:
    if Q is None:
        return True
    x, y = Q
    return f2_mul(y, y) == f2_add(f2_mul(f2_mul(x, x), x), G2_B)


def g2_add(P, Q):
    if P is None:
        return Q
    if Q is None:
        return P
    (x1, y1), (x2, y2) = P, Q
    if x1 == x2:
        if f2_add(y1, y2) == (0, 0):
            return None
        lam = f2_mul(f2_mul((3, 0), f2_mul(x1, x1)), f2_inv(f2_add(y1, y1)))
    else:
        lam = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(lam, lam), x1), x2)
    return (x3, f2_sub(f2_mul(lam, f2_sub(x1, x3)), y1))


def g2_mul(P, k: int):
    acc, add = None, P
    while k:
        if k & 1:
            acc = g2_add(acc, add)
        add = g2_add(add, add)
        k >>= 1
    return acc


def g2_to_limbs(Q):
    """G2Affine as 16 u64 limbs (x.c0, x.c1, y.c0, y.c1, Montgomery): the 128 bytes of g2 / s_g2 in a RawBytes params file."""
    if Q is None:
        return [0] * 16
    out = []
    for c in (Q[0][0], Q[0][1], Q[1][0], Q[1][1]):
        out += mont_limbs(c, P_MOD)
    return out


def g2_from_evm_words(words):
    """(x.c1, x.c0, y.c1, y.c0) big-endian words of a pairing-precompile input -> ((x.c0, x.c1), (y.c0, y.c1))"""
    w = [int(x, 16) if isinstance(x, str) else int(x) for x in words]
    return ((w[1], w[0]), (w[3], w[2]))
