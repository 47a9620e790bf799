"""
oracle/keccak.py -- TEST INFRASTRUCTURE: Keccak-256 (the original padding 0x01, not NIST SHA-3's 0x06 -- Python's hashlib has only the latter), for the transcript the reference's
layer 6 is proved with: snark-verifier's `EvmTranscript` [EXT-recalled snark-verifier system/halo2/transcript/evm.rs], which the released EVM verifier spells out instruction by
instruction [REF release-v0.13.1/evm_verifier.yul:66-100: mstore(0x0, digest); 25 instance words; keccak256(0x0, 896); mod f_q; keccak256(0x3c0, 96); mstore8(.., 1); keccak256(.., 33)].
Pinned by the empty-string and "abc" vectors and, decisively, by the released bundle proof verifying (tests/test_plonk_protocol.py::test_released_bundle_evm_proof_verifies).
"""
_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
       0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
       0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1


def _rol(x, n):
    return ((x << n) | (x >> (64 - n))) & _M64 if n else x


def _f1600(A):
    for rnd in range(24):
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        D = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ D[x] for y in range(5)] for x in range(5)]
        B = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                B[y][(2 * x + 3 * y) % 5] = _rol(A[x][y], _ROT[x][y])
        A = [[B[x][y] ^ ((~B[(x + 1) % 5][y]) & B[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        A[0][0] ^= _RC[rnd]
    return A


def keccak256(data: bytes) -> bytes:
    rate = 136
    p = bytearray(data)
    p.append(0x01)
    while len(p) % rate:
        p.append(0)
    p[-1] |= 0x80
    A = [[0] * 5 for _ in range(5)]
    for off in range(0, len(p), rate):
        for i in range(rate // 8):
            A[i % 5][i // 5] ^= int.from_bytes(p[off + 8 * i:off + 8 * i + 8], "little")
        A = _f1600(A)
    return b"".join(A[i % 5][i // 5].to_bytes(8, "little") for i in range(4))
