"""
oracle/pairing.py -- TEST INFRASTRUCTURE: the BN254 optimal ate pairing in plain Python integers, so that the CPU restatement of the verifier (oracle/plonk.py) can check a
proof the way the reference's verifiers do -- e(lhs, G2) == e(W', [s]G2) with the [s]G2 of the reference's own SRS, which the released EVM verifier carries as four words
[REF release-v0.13.1/evm_verifier.yul:1230-1239] (tests/golden/kat.json "yul") -- instead of with a known trapdoor.  That is what lets the REFERENCE'S RELEASED PROOFS
(tests/golden/kat.json chunk_proof / batch_proof) pin the restatement: they were made with an SRS whose trapdoor nobody knows.

Construction (the textbook one; nothing here is performance code):
  Fp12 = Fp[w] / (w^12 - 18 w^6 + 82): with u^2 = -1 and xi = 9 + u, w^6 = xi, so u = w^6 - 9 and Fp2 embeds by  a + b u -> (a - 9 b) + b w^6
  twist  psi(x, y) = (x w^2, y w^3)  maps E'(Fp2): y^2 = x^3 + 3 / xi  into  E(Fp12): y^2 = x^3 + 3
  Miller loop over 6 t + 2 = 29793968203157093288 with affine line functions, then the two Frobenius steps Q1 = pi(Q), -Q2 = -pi^2(Q), then f^((p^12 - 1) / r).
Only tests/ and this directory's verifier import it.
"""
from . import pyref

P = pyref.P_MOD
R = pyref.R_MOD
ATE_LOOP = 29793968203157093288          # 6 t + 2, t = 4965661367192848881
LOG_ATE = 63
_MOD = (82, 0, 0, 0, 0, 0, -18, 0, 0, 0, 0, 0)   # w^12 = 18 w^6 - 82


class F12:
    __slots__ = ("c",)

    def __init__(self, c):
        self.c = tuple(int(x) % P for x in c)

    @staticmethod
    def one():
        return F12((1,) + (0,) * 11)

    @staticmethod
    def zero():
        return F12((0,) * 12)

    @staticmethod
    def of_fp(a):
        return F12((a,) + (0,) * 11)

    @staticmethod
    def of_fp2(a):                      # a = (c0, c1) = c0 + c1 u
        return F12((a[0] - 9 * a[1], 0, 0, 0, 0, 0, a[1], 0, 0, 0, 0, 0))

    def __eq__(self, o):
        return self.c == o.c

    def __add__(self, o):
        return F12([a + b for a, b in zip(self.c, o.c)])

    def __sub__(self, o):
        return F12([a - b for a, b in zip(self.c, o.c)])

    def __neg__(self):
        return F12([-a for a in self.c])

    def __mul__(self, o):
        if isinstance(o, int):
            return F12([a * o for a in self.c])
        t = [0] * 23
        for i, a in enumerate(self.c):
            if a:
                for j, b in enumerate(o.c):
                    t[i + j] += a * b
        for i in range(22, 11, -1):     # w^i = w^(i-12) (18 w^6 - 82)
            top = t[i]
            if top:
                t[i - 6] += 18 * top
                t[i - 12] -= 82 * top
        return F12(t[:12])

    def is_zero(self):
        return not any(self.c)

    def inv(self):
        """extended Euclid in Fp[w] against the modulus polynomial"""
        lm, hm = [1] + [0] * 12, [0] * 13
        low, high = list(self.c) + [0], [x % P for x in _MOD] + [1]
        deg = lambda p_: max((i for i, x in enumerate(p_) if x), default=0)
        while deg(low):
            r = _poly_rounded_div(high, low)
            r += [0] * (13 - len(r))
            nm, new = list(hm), list(high)
            for i in range(13):
                for j in range(13 - i):
                    nm[i + j] -= lm[i] * r[j]
                    new[i + j] -= low[i] * r[j]
            nm = [x % P for x in nm]
            new = [x % P for x in new]
            lm, low, hm, high = nm, new, lm, low
        i0 = pow(low[0], P - 2, P)
        return F12([x * i0 for x in lm[:12]])

    def __truediv__(self, o):
        return self * o.inv()

    def __pow__(self, e):
        out, b = F12.one(), self
        while e:
            if e & 1:
                out = out * b
            b = b * b
            e >>= 1
        return out


def _poly_rounded_div(a, b):
    dega = max((i for i, x in enumerate(a) if x), default=0)
    degb = max((i for i, x in enumerate(b) if x), default=0)
    temp = list(a)
    o = [0] * len(a)
    binv = pow(b[degb], P - 2, P)
    for i in range(dega - degb, -1, -1):
        q = temp[degb + i] * binv % P
        o[i] = q
        for c in range(degb + 1):
            temp[c + i] = (temp[c + i] - q * b[c]) % P
    return o[:max((i for i, x in enumerate(o) if x), default=0) + 1]


_W = F12((0, 1) + (0,) * 10)
_W2, _W3 = _W * _W, _W * _W * _W


def twist(Q):
    """E'(Fp2) -> E(Fp12); Q = ((x0, x1), (y0, y1)) affine, None = infinity"""
    if Q is None:
        return None
    return (F12.of_fp2(Q[0]) * _W2, F12.of_fp2(Q[1]) * _W3)


def _double(Pt):
    x, y = Pt
    m = (x * x * 3) / (y * 2)
    nx = m * m - x * 2
    return (nx, m * (x - nx) - y)


def _add(P1, P2):
    if P1 is None or P2 is None:
        return P1 if P2 is None else P2
    x1, y1 = P1
    x2, y2 = P2
    if x1 == x2:
        return _double(P1) if y1 == y2 else None
    m = (y2 - y1) / (x2 - x1)
    nx = m * m - x1 - x2
    return (nx, m * (x1 - nx) - y1)


def _line(P1, P2, T):
    """the line through P1 and P2 (the tangent when they coincide), evaluated at T"""
    x1, y1 = P1
    x2, y2 = P2
    xt, yt = T
    if not (x1 == x2):
        m = (y2 - y1) / (x2 - x1)
        return m * (xt - x1) - (yt - y1)
    if y1 == y2:
        m = (x1 * x1 * 3) / (y1 * 2)
        return m * (xt - x1) - (yt - y1)
    return xt - x1


def miller_loop(Q, Pt):
    """Q in G2 (affine over Fp2), Pt in G1 (affine ints); the value BEFORE the final exponentiation (so products of several pairings share one)"""
    if Q is None or Pt is None:
        return F12.one()
    Q12 = twist(Q)
    P12 = (F12.of_fp(Pt[0]), F12.of_fp(Pt[1]))
    Rp = Q12
    f = F12.one()
    for i in range(LOG_ATE, -1, -1):
        f = f * f * _line(Rp, Rp, P12)
        Rp = _double(Rp)
        if ATE_LOOP & (1 << i):
            f = f * _line(Rp, Q12, P12)
            Rp = _add(Rp, Q12)
    Q1 = (Q12[0] ** P, Q12[1] ** P)
    nQ2 = (Q1[0] ** P, -(Q1[1] ** P))
    f = f * _line(Rp, Q1, P12)
    Rp = _add(Rp, Q1)
    f = f * _line(Rp, nQ2, P12)
    return f


FINAL_EXP = (P ** 12 - 1) // R


def final_exponentiation(f):
    return f ** FINAL_EXP


def pairing(Q, Pt):
    return final_exponentiation(miller_loop(Q, Pt))


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 for pairs (P_i in G1, Q_i in G2): what the EVM precompile at address 8 answers"""
    f = F12.one()
    for Pt, Q in pairs:
        f = f * miller_loop(Q, Pt)
    return final_exponentiation(f) == F12.one()
