"""
oracle/poseidon.py -- TEST INFRASTRUCTURE: the Poseidon sponge the reference's inner proofs hash their transcript with, restated from the published construction.

The reference's layers 0-5 are proved with snark-verifier-sdk's `PoseidonTranscript<NativeLoader, _>` (the next layer verifies them in-circuit; only layer 6 uses Keccak)
[REF prover call chain: integration/src/prove.rs:30-43 -> prover::ChunkProver::gen_chunk_proof -> snark-verifier-sdk gen_snark_shplonk, EXT-recalled].  Neither that crate nor
its `poseidon` dependency [REF Cargo.lock:2927-2929 poseidon@5787dd3] is in the checkout, so everything below is the PUBLISHED algorithm:

  * round constants and the MDS matrix from the Grain LFSR of the Poseidon paper's reference script (generate_parameters_grain.sage): 80-bit state seeded with
    field type (2 bits = 1) | s-box (4 bits = 0) | field bits (12) | t (12) | R_F (10) | R_P (10) | thirty 1s, 160 warm-up clocks, self-shrinking output; constants by rejection
    sampling of 254-bit draws, the Cauchy matrix 1 / (x_i + y_j) from 2 t draws reduced mod r;
  * the permutation: R_F / 2 full rounds, R_P partial rounds (s-box x^5 on the first word only), R_F / 2 full rounds; each round = add constants, s-box, multiply by the matrix
    (the poseidon crate runs the optimised schedule with sparse matrices, which computes the same function);
  * the sponge of snark-verifier's util/hash/poseidon.rs: state [2^64, 0, ...]; `update` only buffers; `squeeze` absorbs the buffer RATE words at a time into words 1.., adds 1 to
    the word after a short chunk, runs one more permutation on an empty chunk when the buffer length was a multiple of RATE, and answers word 1.  The state carries over.
  * parameters of the SDK's transcript: T = 5, RATE = 4, R_F = 8, R_P = 60.

How this restatement is pinned (tests/test_plonk_protocol.py):
  1. T = 3, R_F = 8, R_P = 57 reproduces the published test vector poseidon([1, 2]) = 0x115cc0f5...189a of circomlib (same Grain script);
  2. decisive: with T = 5 / 8 / 60 the verifier of oracle/plonk.py ACCEPTS the reference's own released chunk proof and batch proof (tests/golden/kat.json, bytes straight from
     [REF integration/tests/test_data/full_proof_1.json, full_proof_batch_agg_1.json]) under a real pairing check -- every challenge of those proofs went through this sponge, and
     one wrong constant or one wrong padding rule breaks the pairing equation.
Only tests/, bench.py's checker and __graft_entry__.smoke() import this file.  The product's transcript is include/mi355zk_transcript.hpp (C++), compared word for word with this one.
"""
from functools import lru_cache

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
FIELD_BITS = 254


def _grain(t: int, rf: int, rp: int):
    def bits(v, n):
        return [(v >> (n - 1 - i)) & 1 for i in range(n)]
    s = bits(1, 2) + bits(0, 4) + bits(FIELD_BITS, 12) + bits(t, 12) + bits(rf, 10) + bits(rp, 10) + [1] * 30

    def clock():
        nb = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        s.pop(0)
        s.append(nb)
        return nb
    for _ in range(160):
        clock()
    while True:                       # self-shrinking: of each pair of bits, a leading 1 lets the second one through
        nb = clock()
        while nb == 0:
            clock()
            nb = clock()
        yield clock()


@lru_cache(maxsize=None)
def parameters(t: int, rf: int, rp: int):
    """(round constants: (rf + rp) * t field elements, row-major by round; the t x t MDS matrix)"""
    g = _grain(t, rf, rp)

    def draw():
        v = 0
        for _ in range(FIELD_BITS):
            v = (v << 1) | next(g)
        return v
    rc = []
    while len(rc) < (rf + rp) * t:
        v = draw()
        if v < R:
            rc.append(v)
    xs = [draw() % R for _ in range(t)]
    ys = [draw() % R for _ in range(t)]
    mds = tuple(tuple(pow(xs[i] + ys[j], R - 2, R) for j in range(t)) for i in range(t))
    return tuple(rc), mds


def permute(state, t: int, rf: int, rp: int):
    rc, mds = parameters(t, rf, rp)
    st = list(state)
    for r in range(rf + rp):
        st = [(a + rc[r * t + i]) % R for i, a in enumerate(st)]
        if r < rf // 2 or r >= rf // 2 + rp:
            st = [pow(a, 5, R) for a in st]
        else:
            st[0] = pow(st[0], 5, R)
        st = [sum(mds[i][j] * st[j] for j in range(t)) % R for i in range(t)]
    return st


class Sponge:
    """snark-verifier's `Poseidon<F, L, T, RATE>` (update / squeeze)"""

    def __init__(self, t: int = 5, rf: int = 8, rp: int = 60):
        self.t, self.rate, self.rf, self.rp = t, t - 1, rf, rp
        self.state = [1 << 64] + [0] * (t - 1)
        self.buf = []

    def update(self, words):
        self.buf += [int(x) % R for x in words]

    def _absorb_and_permute(self, chunk):
        st = self.state
        for i, x in enumerate(chunk):
            st[1 + i] = (st[1 + i] + x) % R
        if len(chunk) < self.rate:
            st[len(chunk) + 1] = (st[len(chunk) + 1] + 1) % R
        self.state = permute(st, self.t, self.rf, self.rp)

    def squeeze(self) -> int:
        buf, self.buf = self.buf, []
        for i in range(0, len(buf), self.rate):
            self._absorb_and_permute(buf[i:i + self.rate])
        if len(buf) % self.rate == 0:
            self._absorb_and_permute([])
        return self.state[1]


def hash_circomlib(inputs):
    """circomlib's fixed-width hash (state [0, inputs...], answer word 0): only here to check the constants against a published vector"""
    t = len(inputs) + 1
    assert t == 3
    return permute([0] + [int(x) % R for x in inputs], t, 8, 57)[0]
