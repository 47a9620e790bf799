"""helpers shared by the -m gpu tests (TEST INFRASTRUCTURE)."""
import numpy as np

from oracle import cref, pyref

R = pyref.R_MOD


def rand_fr(rng, n, full=True):
    """n random field elements as Montgomery limbs [n,4] (uniform below 2^252, plus a few extreme values)."""
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    if full and n >= 4:
        a[0] = cref.fr_mont(R - 1); a[1] = cref.fr_mont(1); a[2] = cref.fr_mont(0); a[3] = cref.fr_mont((1 << 253) + 12345)
    return a


def rand_points(rng, n):
    """n random curve points [n,8] through the oracle (slow: use for n <= ~4096)."""
    G = cref.g1_generator()
    sc = rand_fr(rng, n, full=False)
    jac = np.stack([cref.g1_mul(G, sc[i]) for i in range(n)])
    return cref.g1_to_affine(jac)


def affine_of(g1):
    """12-limb normalised Jacobian from the library -> 8-limb affine, checking the normalisation contract."""
    g1 = np.asarray(g1)
    one_q = np.array(pyref.to_limbs(pyref.MONT_R % pyref.P_MOD), dtype=np.uint64)
    if (g1[8:] == 0).all():
        assert (g1 == 0).all(), "identity must be returned as all-zero"
        return np.zeros(8, dtype=np.uint64)
    assert (g1[8:] == one_q).all(), "result must be normalised: z == R (Montgomery one)"
    return g1[:8].copy()


def oracle_affine(jac):
    return cref.g1_to_affine(jac)
