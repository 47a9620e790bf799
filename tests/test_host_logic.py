"""CPU-only: the host side above the C-ABI (halo2 mirror constants, shard arithmetic) against the fixtures and the oracle."""
import numpy as np
import pytest

import __graft_entry__ as ge
from oracle import cref, pyref


@pytest.fixture(scope="module")
def zk():
    return ge.load_package()


@pytest.mark.parametrize("which,k", [("chunk_protocol", 25), ("batch_proof", 26)])
def test_evaluation_domain_constants_match_reference_fixtures(zk, kat, which, k):
    """EvaluationDomain::new derives omega from ROOT_OF_UNITY; the released protocol files store the same value (KAT A1/A2)."""
    pr = kat[which] if which == "chunk_protocol" else kat[which]["protocol"]
    dom = zk.halo2.EvaluationDomain(2, k)
    assert dom.omega.tolist() == pr["domain"]["gen"]
    assert dom.omega_inv.tolist() == pr["domain"]["gen_inv"]
    assert dom.ifft_divisor.tolist() == pr["domain"]["n_inv"]


def test_extended_domain_rule(zk):
    h2 = zk.halo2
    for j, k, want in [(2, 10, 10), (3, 10, 11), (4, 10, 12), (5, 10, 12), (6, 10, 13), (5, 26, 28)]:
        dom = h2.EvaluationDomain(j, k)
        assert dom.extended_k == want
        assert pow(dom._extended_omega, 1 << dom.extended_k, h2.R_MOD) == 1 and pow(dom._extended_omega, 1 << (dom.extended_k - 1), h2.R_MOD) != 1
        assert pow(dom._extended_omega, 1 << (dom.extended_k - k), h2.R_MOD) == dom._omega
    with pytest.raises(AssertionError):
        h2.EvaluationDomain(9, 26)      # would need 2^29 > two-adicity 28
    assert (h2.fr(h2.FR_ZETA) == cref.fr_mont(pyref.FR_ZETA)).all()
    for v in (0, 1, 7, h2.R_MOD - 1, 1 << 200):
        assert (h2.fr(v) == cref.fr_mont(v)).all() and h2.fr_to_int(h2.fr(v)) == v % h2.R_MOD


def test_shard_ranges(zk):
    sr = zk.distributed.shard_range
    for n in (0, 1, 7, 8, 1000, 1 << 26):
        for world in (1, 2, 3, 4, 8):
            cover = []
            for r in range(world):
                lo, hi = sr(n, r, world)
                assert 0 <= lo <= hi <= n
                cover.append((lo, hi))
            assert cover[0][0] == 0 and cover[-1][1] == n
            assert all(cover[i][1] == cover[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cover]
            assert max(sizes) - min(sizes) <= 1


def test_params_file_format_roundtrip(zk, tmp_path):
    """SerdeFormat::RawBytes layout of params{k} and the exact-length rule of prover::load_params."""
    h2 = zk.halo2
    assert h2.params_file_size(26) == 8589934852 and h2.params_file_size(20) == 134217988   # the sizes behind [REF params-sha256sum:1-5]
    k = 3
    g, gl, _, _ = cref.srs_setup(k, cref.fr_mont(0xABCDEF), cref.fr_mont(pyref.omega(k)))
    path = str(tmp_path / "params3")
    g2, s_g2 = bytes(range(128)), bytes(range(128, 256))
    h2.write_params(path, k, g, gl, g2, s_g2)
    k2, a, b, c, d = h2.read_params(path)
    assert k2 == k and (a == g).all() and (b == gl).all() and c == g2 and d == s_g2
    with open(path, "ab") as f:
        f.write(b"\\0")
    with pytest.raises(ValueError):
        h2.read_params(path)


@pytest.mark.parametrize("vk,proto,npts", [("vk_chunk", "chunk_protocol", 7), ("vk_batch_agg", "batch_proof", 9)])
def test_g1_wire_codec_against_released_vkeys(zk, kat, vk, proto, npts):
    """the host mirror's compressed-point codec on the reference's own .vkey bytes and protocol points (KAT A4), and on proof words (A5/A6)."""
    h2 = zk.halo2
    raw = bytes.fromhex(kat[vk])
    pr = kat[proto] if proto == "chunk_protocol" else kat[proto]["protocol"]
    for i in range(npts):
        word = raw[8 + 32 * i: 40 + 32 * i]
        pt = np.array(pr["preprocessed"][i]["x"] + pr["preprocessed"][i]["y"], dtype=np.uint64)
        assert (h2.g1_from_bytes(word) == pt).all() and h2.g1_to_bytes(pt) == word
        assert (h2.g1_from_bytes(word) == cref.g1_decompress(word)).all()
    proof = bytes.fromhex(kat["chunk_proof"]["proof"])
    for i in list(range(9)) + [26, 27]:
        w = proof[32 * i: 32 * i + 32]
        assert h2.g1_to_bytes(h2.g1_from_bytes(w)) == w and cref.g1_is_on_curve(h2.g1_from_bytes(w))
    assert h2.g1_to_bytes(np.zeros(12, dtype=np.uint64)) == bytes(32) and (h2.g1_from_bytes(bytes(32)) == 0).all()
    with pytest.raises(ValueError):
        h2.g1_from_bytes((4).to_bytes(32, "little"))      # x = 4: 4^3 + 3 = 67 is not a square mod p


def test_signed_digit_recoding_algorithm():
    """The recoding k_msm_digits applies (restated here in integers): take the smaller of k and r - k, cut it into c-bit windows from
    the bottom, turn raw values above 2^(c-1) into negative digits with a carry.  Properties the kernels rely on: W = ceil(255 / c)
    windows always absorb the last carry, |digit| <= 2^(c-1) (bucket index = |digit| - 1 < 2^(c-1)), and the digits reconstruct
    +-k (mod r) with the sign that moved to the point."""
    import random
    R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
    rng = random.Random(2024)

    def recode(k, c):
        flip = (R - k) < k
        v = R - k if flip else k
        W = (255 + c - 1) // c
        half, digits, carry = 1 << (c - 1), [], 0
        for w in range(W):
            raw = ((v >> (w * c)) & ((1 << c) - 1)) + carry
            if raw > half:
                d, carry = raw - (1 << c), 1
            else:
                d, carry = raw, 0
            digits.append(-d if flip else d)
        return digits, carry

    samples = [0, 1, 2, R - 1, R - 2, (R - 1) // 2, (R + 1) // 2, (1 << 253) - 1, (1 << 252), R >> 1, 0x1234567890abcdef1234567890abcdef1234567890abcdef]
    samples += [rng.randrange(R) for _ in range(300)] + [rng.randrange(1 << 64) for _ in range(50)] + [R - rng.randrange(1 << 64) for _ in range(50)]
    for c in range(2, 25):
        for k in samples:
            digits, carry = recode(k, c)
            assert carry == 0, (c, hex(k))
            assert all(abs(d) <= 1 << (c - 1) for d in digits)
            assert sum(d << (w * c) for w, d in enumerate(digits)) % R == k % R, (c, hex(k))
        # small negative values use ONE window once c covers them
        if c >= 8:
            digits, _ = recode(R - 100, c)
            assert digits[0] == -100 and all(d == 0 for d in digits[1:])


def test_accumulate_segment_rule_covers_every_entry():
    """msm_seg_eff (msm.hpp), restated in integers: the accumulate launch is sized for the worst case (threads = ceil(emax / seg_max) rounded to
    workgroups), and every kernel derives the segment from the actual entry count as clamp(ceil(total / (fill % of the threads)), seg_min,
    seg_max).  Whatever the count, threads * segment must cover it, and a full column must get the worst-case segment back."""
    import random

    def seg_eff(total, threads, seg_max, seg_min, fill_pct):
        target = (threads * min(fill_pct, 100) + 99) // 100          # the kernel clamps the percentage: more than the launched threads cannot take entries
        sg = (total + target - 1) // max(target, 1)
        return min(max(sg, seg_min), seg_max)

    rng = random.Random(7)
    for _ in range(20000):
        emax = rng.choice([rng.randrange(1, 1 << 12), rng.randrange(1, 1 << 24), rng.randrange(1, (1 << 32) - 1)])
        want_threads = 256 * 256 * 16
        seg_max = min(max((emax + want_threads - 1) // want_threads, 16), 4096)
        threads = -(-(-(-emax // seg_max)) // 256) * 256
        seg_min = min(rng.choice([1, 16, 64, 256, 4096]), seg_max)
        fill = rng.choice([2, 20, 40, 60, 100, 126])
        total = rng.choice([0, 1, emax, emax // 7, rng.randrange(0, emax + 1)])
        sg = seg_eff(total, threads, seg_max, seg_min, fill)
        assert seg_min <= sg <= seg_max
        assert threads * sg >= total, (emax, total, threads, sg)
        if total == emax and fill <= 100:
            assert sg == seg_max or sg * threads >= emax      # nothing is lost for a column in which every digit is non-zero


def test_build_is_content_hashed_per_translation_unit():
    """scroll-prover_amd/build.py decides staleness by content (sha256 of every file a unit includes + the flags), so a copied tree -- the GPU
    box's snapshot -- never rebuilds what was built here, and a header change rebuilds exactly the units that include it."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_b", os.path.join(root, "scroll-prover_amd", "build.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    b.build()
    assert not b.needs_build()
    assert b.needs_build(extra_flags=["-DSOMETHING=1"])                       # other flags = another build
    h = {u: b._unit_hash(u, []) for u in b.UNITS}
    assert len(set(h.values())) == len(b.UNITS)
    closure = {}
    b._closure(os.path.join(b.CSRC, "lib_core.hip"), closure)
    names = {os.path.basename(p) for p in closure}
    assert "lib_common.hpp" in names and "mi355zk.h" in names and "msm.hpp" not in names and "ntt29.hpp" not in names   # lib_core launches no kernel
    closure = {}
    b._closure(os.path.join(b.CSRC, "lib_ntt.hip"), closure)
    assert "ntt29.hpp" in {os.path.basename(p) for p in closure} and "msm.hpp" not in {os.path.basename(p) for p in closure}


def test_host_compact_nonzero_needs_no_device():
    """the host half of the sparse upload (include/mi355zk.h mi355_host_compact_nonzero): indices and values of the non-zero cells in order, any thread count, ragged sizes"""
    import numpy as np
    import __graft_entry__ as ge
    h2 = ge.load_package().halo2
    rng = np.random.default_rng(5)
    for n in (0, 1, 1000, (1 << 17) + 13):
        col = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
        col[rng.random(n) < 0.7] = 0
        want = np.nonzero(col.any(axis=1))[0].astype(np.uint32)
        for threads in (1, 3, 16):
            idx, vals = h2.compact_nonzero(col, threads)
            assert (idx == want).all() and (vals == col[want]).all()
