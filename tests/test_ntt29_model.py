"""The NTT pass (ezkl_amd/csrc/ntt.hip, ntt_pass_kernel) as an integer model, tools/ntt29_model.py: the lazily reduced radix-2^29 DIT
butterfly with every 32- / 64-bit register checked, and the pass plan with the kernel's own index arithmetic (tiles, column bases, digit
reversal, inter-pass tables, the twisted first pass of the coset-major transform), against the plain DFT over Fr.  Host logic only: the
kernel itself is compared with the oracle by tests/test_gpu_ntt.py on the GPU."""
import random
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import ntt29_model as M


def test_column_transforms_stay_in_range_and_match_the_dft():
    worst = M.selftest(seed=3, log_rs=(1, 2, 3, 4, 6, 7, 8))
    # the bound the kernel's comment states: 46p after the two product-free stages (a 256-bit input, 8p and 32p borrowed), 4p per later stage
    assert worst["max_value_over_p"] < 82
    # ... and without the product-free stage-2 butterflies (EZKL_NTT_SKIP_UNIT2=0): 4p per stage on top of a 256-bit input plus the first stage's 8p
    M.SKIP_UNIT2 = False
    try:
        assert M.selftest(seed=3, log_rs=(3, 6, 8))["max_value_over_p"] < 54
    finally:
        M.SKIP_UNIT2 = True
    # three stages between carry passes would also stay inside 32-bit limbs (the kernel uses two: register pressure)
    assert M.selftest(seed=4, log_rs=(3, 6, 7), group=3)["max_value_over_p"] < 82


def test_constants_are_the_generated_ones():
    src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ezkl_amd", "csrc", "montmul29_gen.hpp")).read()
    fr = src[src.index("struct Fr29C"):]
    def row(tag):
        line = [l for l in fr.splitlines() if tag in l][0]
        return [int(x.rstrip("u"), 16) for x in line[line.index("{") + 1:line.index("}")].split(", ")]
    assert row("// 2^261 - 1 p") == M.CSUB_P
    assert row("// 4 p") == M.SUBC[4] and row("// 8 p") == M.SUBC[8] and row("// 32 p") == M.SUBC[32]
    assert row("P[9]") == M.limbs29(M.P)


GEN = pow(7, (M.P - 1) >> 28, M.P)


def _dft(a, w, k):
    return sum(x * pow(w, i * k, M.P) for i, x in enumerate(a)) % M.P


def test_two_pass_plan_indexing_inverse_scale_and_padding():
    """2^13 = passes of 7 + 6 bits (unequal radices, the k1-major last pass): inverse transform with its 1 / n product in the last pass, and
    a forward transform of an input zero-padded from 2^11"""
    rnd = random.Random(11)
    log_n = 13
    n = 1 << log_n
    assert M.plan_radices(log_n) == [7, 6]
    w = pow(GEN, 1 << (28 - log_n), M.P)
    winv, ninv = pow(w, -1, M.P), pow(n, -1, M.P)
    a = [rnd.randrange(M.P) for _ in range(n)]
    got = M.transform(a, log_n, winv, inverse_scale=True)
    for k in rnd.sample(range(n), 4):
        assert got[k] == _dft(a, winv, k) * ninv % M.P
    got = M.transform(a, log_n, w, in_log_len=11)
    for k in rnd.sample(range(n), 4):
        assert got[k] == _dft(a[:1 << 11], w, k)


def test_three_pass_plan_indexing():
    """a middle pass and a three-digit reversal: 2^13 as 5 + 4 + 4 (EZKL_NTT_MAXR=6)"""
    rnd = random.Random(12)
    log_n = 13
    n = 1 << log_n
    w = pow(GEN, 1 << (28 - log_n), M.P)
    a = [rnd.randrange(M.P) for _ in range(n)]
    orig = M.plan_radices
    M.plan_radices = lambda ln, maxr=6: orig(ln, 6)
    try:
        assert M.plan_radices(log_n) == [5, 4, 4]
        got = M.transform(a, log_n, w)
    finally:
        M.plan_radices = orig
    for k in rnd.sample(range(n), 4):
        assert got[k] == _dft(a, w, k)


def test_coset_major_transform_twisted_first_pass():
    """coefficients -> the two cosets of a 2^12 -> 2^13 extended domain: coset b at zeta w_ext^b <w_n>, the scaling absorbed by the first
    pass's stage twiddles and its inter-pass table (no product for it)"""
    rnd = random.Random(13)
    log_n, log_e = 12, 1
    n = 1 << log_n
    w_ext = pow(GEN, 1 << (28 - log_n - log_e), M.P)
    w_n = pow(w_ext, 2, M.P)
    zeta = 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23
    a = [rnd.randrange(M.P) for _ in range(n)]
    outs = M.coset_transform(a, log_n, log_e, w_ext, zeta)
    for b in range(2):
        for j in rnd.sample(range(n), 3):
            x = zeta * pow(w_ext, b, M.P) * pow(w_n, j, M.P) % M.P
            want = 0
            for c in reversed(a):
                want = (want * x + c) % M.P
            assert outs[b][j] == want


def test_natural_order_coset_transforms_round_trip():
    """EvaluationDomain::coeff_to_extended / extended_to_coeff in natural order (ezkl_hip_coset_ntt_*: zeta^(i mod 3) on the way in, zeta^-(i mod 3) / n
    on the way out, fused into the first / last pass): 2^10 coefficients -> 2^12 evaluations on zeta <w_ext> -> the coefficients again"""
    rnd = random.Random(14)
    log_n, log_ext = 10, 12
    w_ext = pow(GEN, 1 << (28 - log_ext), M.P)
    zeta = 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23
    a = [rnd.randrange(M.P) for _ in range(1 << log_n)]
    padded = a + [rnd.randrange(M.P) for _ in range((1 << log_ext) - (1 << log_n))]        # what lies beyond 2^in_log_len is read as zero
    ev = M.transform(padded, log_ext, w_ext, in_log_len=log_n, coset_mode=1, zeta=zeta)
    for j in rnd.sample(range(1 << log_ext), 3):
        x = zeta * pow(w_ext, j, M.P) % M.P
        want = 0
        for c in reversed(a):
            want = (want * x + c) % M.P
        assert ev[j] == want
    back = M.transform(ev, log_ext, pow(w_ext, -1, M.P), coset_mode=2, zeta=zeta)
    assert back[:1 << log_n] == a and not any(back[1 << log_n:])
