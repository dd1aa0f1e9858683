"""Interpret the GENERATED inline-asm Montgomery products (ezkl_amd/csrc/montmul29_gen.hpp) instruction by instruction on the CPU:
every v_mad_u64_u32 is checked for 64-bit overflow, the result for congruence and for its limb / value bounds.  No GPU needed -- this
is what pins the generator (tools/gen_montmul29.py) for forms that cannot be compared with an older kernel (the two-product form
mont_mul2add29_fq of round 5: r = (a b + c d) / 2^261 with one reduction)."""
import os
import random
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "ezkl_amd", "csrc", "montmul29_gen.hpp")
FQ = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
FR = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
M29 = (1 << 29) - 1


def asm_of(name):
    src = open(HDR).read()
    i = src.index("void %s(" % name)
    j = src.index("\n}", i)
    return re.findall(r'"([^"\\]+)\\n\\t"', src[i:j])


class Overflow(Exception):
    pass


def run(lines, operands):
    """operands: {'a': [9 limbs], 'b': ...}; returns r[0..8].  Registers: vN / sN by name, %[xN] by operand name."""
    reg = {}
    for n, v in operands.items():
        for i, x in enumerate(v):
            reg["%%[%s%d]" % (n, i)] = x

    def rd(tok):
        tok = tok.strip()
        if tok == "0": return 0
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m: return reg.get("v" + m.group(1), 0) | (reg.get("v" + m.group(2), 0) << 32)
        if tok.startswith("0x"): return int(tok, 16)
        if tok.isdigit(): return int(tok)
        return reg[tok]

    def wr(tok, val, wide=False):
        tok = tok.strip()
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
        if m:
            reg["v" + m.group(1)] = val & 0xffffffff
            reg["v" + m.group(2)] = (val >> 32) & 0xffffffff
        else:
            reg[tok] = val & 0xffffffff

    for ln in lines:
        op, rest = ln.split(None, 1)
        args = [a.strip() for a in re.split(r",\s*(?![^\[]*\])", rest)]
        if op == "s_mov_b32": wr(args[0], rd(args[1]))
        elif op == "v_mov_b32": wr(args[0], rd(args[1]))
        elif op == "v_mad_u64_u32":
            x, y = rd(args[2]), rd(args[3])
            assert x < (1 << 32) and y < (1 << 32)
            t = x * y + rd(args[4])
            if t >> 64: raise Overflow(ln)
            wr(args[0], t)
        elif op == "v_mul_lo_u32": wr(args[0], (rd(args[1]) * rd(args[2])) & 0xffffffff)
        elif op == "v_and_b32": wr(args[0], rd(args[1]) & rd(args[2]))
        elif op == "v_lshrrev_b64": wr(args[0], rd(args[2]) >> rd(args[1]))
        elif op == "v_lshlrev_b32": wr(args[0], (rd(args[2]) << rd(args[1])) & 0xffffffff)
        elif op == "v_lshl_add_u64":
            t = (rd(args[1]) << rd(args[2])) + rd(args[3])
            if t >> 64: raise Overflow(ln)
            wr(args[0], t)
        else:
            raise AssertionError("unknown instruction " + ln)
    return [reg["%%[r%d]" % i] for i in range(9)]


def value(v):
    return sum(x << (29 * i) for i, x in enumerate(v))


def loose_limbs(rng, mod, vmax_p, units):
    """a value < vmax_p * mod written with limbs up to `units` * 2^29 (limb 8 holds the rest): what add / sub / neg hand to a product"""
    x = rng.randrange(vmax_p * mod)
    v = [(x >> (29 * i)) & M29 for i in range(8)] + [x >> 232]
    for i in range(8):                        # borrow from the limb above to fatten this one
        k = rng.randrange(units)
        k = min(k, v[i + 1] >> 0 if i + 1 < 8 else v[8])
        if k and v[i + 1] >= k:
            v[i + 1] -= k
            v[i] += k << 29
    assert value(v) == x and all(l < units * (1 << 29) + (1 << 29) for l in v[:8])
    return v, x


def test_plain_product_and_square():
    rng = random.Random(1)
    rinv = pow(1 << 261, -1, FQ)
    mul, sqr = asm_of("mont_mul29_fq"), asm_of("mont_sqr29_fq")
    for _ in range(40):
        (a, av), (b, bv) = loose_limbs(rng, FQ, 20, 3), loose_limbs(rng, FQ, 20, 1)
        r = run(mul, {"a": a, "b": b})
        assert value(r) % FQ == av * bv * rinv % FQ and all(x <= M29 for x in r[:8]) and value(r) < (400 // 169 + 2) * FQ
        r = run(sqr, {"a": a}) if all(x < (1 << 31) for x in a) else None
    a, av = loose_limbs(rng, FQ, 20, 2)
    r = run(sqr, {"a": a})
    assert value(r) % FQ == av * av * rinv % FQ


def test_two_product_form_one_reduction():
    """r = (a b + c d) / 2^261: the operand shapes of Y3 = R (Q - X3) + (K p - S1) PPP in curve29.hpp -- normalized x loose(3) plus
    loose(2) x normalized, values below 18p, 18p, 16p, 2p -- never overflow a column and land below 4p"""
    rng = random.Random(2)
    rinv = pow(1 << 261, -1, FQ)
    lines = asm_of("mont_mul2add29_fq")
    assert sum(l.startswith("v_mad_u64_u32") for l in lines) == 243
    worst = 0
    for it in range(60):
        (a, av), (b, bv) = loose_limbs(rng, FQ, 18, 1), loose_limbs(rng, FQ, 18, 3)
        (c, cv), (d, dv) = loose_limbs(rng, FQ, 16, 2), loose_limbs(rng, FQ, 2, 1)
        if it == 0:                            # the extreme limbs the bound allows
            a = [M29] * 8 + [a[8]]; b = [3 * (1 << 29) - 1] * 8 + [b[8]]; c = [2 * (1 << 29) - 1] * 8 + [c[8]]; d = [M29] * 8 + [d[8]]
            av, bv, cv, dv = value(a), value(b), value(c), value(d)
        r = run(lines, {"a": a, "b": b, "c": c, "d": d})
        assert value(r) % FQ == (av * bv + cv * dv) * rinv % FQ
        assert all(x <= M29 for x in r[:8])
        if it: worst = max(worst, value(r) / FQ)
    assert worst < 4.0


def test_two_product_form_overflows_when_the_bound_is_broken():
    """the interpreter does catch an overflow: loose(3) x loose(3) + loose(3) x loose(3) breaks A B + C D < 6.1"""
    lines = asm_of("mont_mul2add29_fq")
    big = [3 * (1 << 29) - 1] * 8 + [1 << 26]
    try:
        run(lines, {"a": big, "b": big, "c": big, "d": big})
    except Overflow:
        return
    raise AssertionError("expected a 64-bit column overflow")
