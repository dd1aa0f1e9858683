// Keccak-256 Fiat-Shamir transcript in the EVM format ezkl proves with (EvmTranscript of snark-verifier, selected at
// /root/reference/src/execute.rs:1608-1609; layout verified on the reference's proof fixture):
// points are absorbed / written as 32-byte big-endian x || y in standard form, scalars as 32-byte big-endian; a
// challenge is keccak256(buffer [|| 0x01 when the buffer is exactly one earlier digest]) reduced mod r, and the digest
// becomes the new buffer.  Same sponge as ezkl_amd/transcript.py (the two are compared byte for byte in the tests).
#pragma once
#include <vector>
#include "hostfield.hpp"

namespace ezkl_prover {

inline uint64_t rotl64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

inline void keccak_f1600(uint64_t a[25]) {      // a[x + 5 y]
    static const uint64_t RC[24] = {0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808Aull, 0x8000000080008000ull, 0x000000000000808Bull,
                                    0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008Aull, 0x0000000000000088ull,
                                    0x0000000080008009ull, 0x000000008000000Aull, 0x000000008000808Bull, 0x800000000000008Bull, 0x8000000000008089ull,
                                    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800Aull, 0x800000008000000Aull,
                                    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int ROT[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};   // [x][y]
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(a[x + 5 * y] ^ d[x], ROT[x][y]);
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= RC[round];
    }
}
inline std::array<uint8_t, 32> keccak256(const uint8_t* data, size_t len) {
    const size_t rate = 136;
    std::vector<uint8_t> p(data, data + len);
    p.push_back(0x01);
    while (p.size() % rate) p.push_back(0);
    p.back() |= 0x80;
    uint64_t a[25] = {0};
    for (size_t off = 0; off < p.size(); off += rate) {
        for (size_t i = 0; i < rate / 8; i++) {
            uint64_t w = 0;
            for (int b = 0; b < 8; b++) w |= (uint64_t)p[off + 8 * i + b] << (8 * b);
            a[i] ^= w;
        }
        keccak_f1600(a);
    }
    std::array<uint8_t, 32> out;
    for (int i = 0; i < 4; i++)
        for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(a[i] >> (8 * b));
    return out;
}

class EvmTranscript {
  public:
    void common_point(const G1& p) {
        U256 x, y;
        p.canonical(x, y);
        uint8_t b[64];
        to_be32(x, b);
        to_be32(y, b + 32);
        buf_.insert(buf_.end(), b, b + 64);
    }
    void common_scalar(const Fe& s) {
        uint8_t b[32];
        to_be32(s.canonical(), b);
        buf_.insert(buf_.end(), b, b + 32);
    }
    void write_point(const G1& p) {
        common_point(p);
        proof_.insert(proof_.end(), buf_.end() - 64, buf_.end());
    }
    void write_scalar(const Fe& s) {
        common_scalar(s);
        proof_.insert(proof_.end(), buf_.end() - 32, buf_.end());
    }
    Fe squeeze_challenge() {
        if (buf_.size() == 32) buf_.push_back(0x01);
        auto h = keccak256(buf_.data(), buf_.size());
        buf_.assign(h.begin(), h.end());
        return Fe::from_canonical(reduce_fr(from_be32(h.data())));
    }
    const std::vector<uint8_t>& proof() const { return proof_; }

  private:
    std::vector<uint8_t> buf_, proof_;
};

}  // namespace ezkl_prover
