"""Keccak-256 Fiat-Shamir transcript in the EVM format ezkl proves with
(EvmTranscript from snark-verifier, selected at /root/reference/src/execute.rs:1608-1609; proof layout verified on the
reference fixture): points are absorbed / written as 32-byte big-endian x || y in standard form,
scalars as 32-byte big-endian; a challenge is keccak256(buffer [|| 0x01 if the buffer is exactly one earlier digest])
reduced mod r, and the digest becomes the new buffer."""

_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
       0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
       0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
       0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M = (1 << 64) - 1
R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
Q = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47


def _rol(x, n):
    return ((x << n) | (x >> (64 - n))) & _M if n else x


def _f1600(a):
    for rc in _RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    return a


def keccak256(data):
    rate = 136
    p = bytearray(data)
    p.append(0x01)
    while len(p) % rate:
        p.append(0)
    p[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(p), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(p[off + 8 * i: off + 8 * i + 8], "little")
        a = _f1600(a)
    return b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


class EvmTranscript:
    """write side (prover) and read side (verifier) share the sponge logic"""

    def __init__(self, proof=b""):
        self.buf = bytearray()
        self.proof = bytearray()          # prover output
        self._rd, self._in = 0, bytes(proof)

    # --- absorb
    def common_point(self, p):
        x, y = (0, 0) if p is None else p
        self.buf += x.to_bytes(32, "big") + y.to_bytes(32, "big")

    def common_scalar(self, s):
        self.buf += (s % R).to_bytes(32, "big")

    # --- prover
    def write_point(self, p):
        self.common_point(p)
        x, y = (0, 0) if p is None else p
        self.proof += x.to_bytes(32, "big") + y.to_bytes(32, "big")

    def write_scalar(self, s):
        self.common_scalar(s)
        self.proof += (s % R).to_bytes(32, "big")

    # --- verifier
    def read_point(self):
        b = self._in[self._rd:self._rd + 64]
        if len(b) != 64:
            raise ValueError("proof too short")
        self._rd += 64
        p = (int.from_bytes(b[:32], "big"), int.from_bytes(b[32:], "big"))
        if p[0] >= Q or p[1] >= Q:
            raise ValueError("non-canonical point")
        p = None if p == (0, 0) else p
        self.common_point(p)
        return p

    def read_scalar(self):
        b = self._in[self._rd:self._rd + 32]
        if len(b) != 32:
            raise ValueError("proof too short")
        self._rd += 32
        s = int.from_bytes(b, "big")
        if s >= R:
            raise ValueError("non-canonical scalar")
        self.common_scalar(s)
        return s

    def squeeze_challenge(self):
        data = bytes(self.buf) + (b"\x01" if len(self.buf) == 32 else b"")
        h = keccak256(data)
        self.buf = bytearray(h)
        return int.from_bytes(h, "big") % R
