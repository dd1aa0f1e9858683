"""Host-side Fq arithmetic of the MSM's tail (ezkl_amd/csrc/host64.hpp), compiled with g++ and run on the CPU: the binary-Euclid
inversion (round 5: every synchronous MSM ends with one inversion on the host) against the exponentiation it replaced and against
a * a^-1 = 1, on random values and on the edge values."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <stdio.h>
#include <stdlib.h>
#include "host64.hpp"
using namespace ezkl::h64;
int main() {
    unsigned long long s = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    int bad = 0;
    for (int it = 0; it < 3000; it++) {
        fe a = {{rnd(), rnd(), rnd(), rnd() >> 3}};
        if (it == 0) a = ONE;
        if (it == 1) { a = Q; a.v[0] -= 1; }                    // q - 1
        if (it == 2) a = fe{{2, 0, 0, 0}};
        if (it == 3) a = fe{{0, 0, 0, 1ull << 60}};
        if (geq(a, Q)) sub4(a, a, Q);
        if (is_zero(a)) continue;
        const fe x = inv(a), y = inv_pow(a);
        if (!eq(x, y) || !eq(mul(a, x), ONE)) bad++;
    }
    const fe z = {{0, 0, 0, 0}};
    if (!is_zero(inv(z))) bad++;
    printf("bad %d\n", bad);
    return bad != 0;
}
"""


def test_binary_euclid_inversion_matches_the_exponentiation():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write(SRC)
        exe = os.path.join(d, "t")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "ezkl_amd", "csrc"), src, "-o", exe])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.strip() == "bad 0", out.stdout + out.stderr
