#!/usr/bin/env python3
"""Integer model of the radix-2^29 NTT pass (ezkl_amd/csrc/ntt.hip, ntt_pass29_kernel): the lazily reduced decimation-in-time butterfly,
its limb and value bounds, the 8 x 32 <-> 9 x 29 conversions and the final canonicalisation, with every 32- / 64-bit register checked
for overflow.  Runs on the CPU (tests/test_ntt29_model.py): the kernel itself needs the GPU.

Representation (field29.hpp): 9 limbs, value = sum v[i] 2^(29 i); data stay in the 2^256 Montgomery domain of the files, twiddles are
kept in the 2^261 domain, so mont(a, w) = a w / 2^261 leaves the data's domain alone."""
import random

P = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
M29 = (1 << 29) - 1
R261 = 1 << 261
PINV = (-pow(P, -1, 1 << 29)) % (1 << 29)


def limbs29(x, n=9):
    return [(x >> (29 * i)) & M29 if i < n - 1 else (x >> (29 * i)) for i in range(n)]


def value(v):
    return sum(x << (29 * i) for i, x in enumerate(v))


def u32(x):
    assert 0 <= x < (1 << 32), "32-bit register overflow: %x" % x
    return x


def u64(x):
    assert 0 <= x < (1 << 64), "64-bit accumulator overflow"
    return x


def unpack(w):
    """Field29::unpack: any 256-bit value as 8 words -> normalized limbs"""
    assert 0 <= w < (1 << 256)
    return limbs29(w)


def pack(v):
    """Field29::pack: normalized limbs of a value < 2^256 -> the 256-bit number"""
    assert all(x <= M29 for x in v[:8])
    x = value(v)
    assert x < (1 << 256)
    return x


def normalize(v):
    r = list(v)
    for i in range(8):
        assert r[i] < (1 << 32) - 8
        r[i + 1] = u32(r[i + 1] + (r[i] >> 29))
        r[i] &= M29
    return r


def add(a, b):
    return [u32(x + y) for x, y in zip(a, b)]


def subc(K):
    k = limbs29(K * P)
    c = [k[0] + (1 << 29)] + [k[i] + (1 << 29) - 1 for i in range(1, 8)] + [k[8] - 1]
    assert value(c) == K * P
    return c


SUBC = {K: subc(K) for K in (2, 4, 8, 16, 32)}
SKIP_UNIT2 = True          # stage 2 of a plain transform: the twiddle of every other butterfly is w_4^0 = 1 -- no product (ntt.hip, round 5)


def sub(a, b, K):
    """a - b + K p; b normalized (limbs 0..7 < 2^29) and < (K - 1) p"""
    assert all(x <= M29 for x in b[:8]) and value(b) < (K - 1) * P
    c = SUBC[K]
    out = []
    for i in range(9):
        d = c[i] - b[i]
        assert d >= 0, "limb difference went negative"
        out.append(u32(a[i] + d))
    return out


def mont_mul(a, b):
    """mont_mul29_fr: a b / 2^261 mod p, columns in one 64-bit accumulator (tools/gen_montmul29.py)"""
    p = limbs29(P)
    m = [0] * 9
    r = [0] * 9
    acc = 0
    for k in range(17):
        for i in range(max(0, k - 8), min(k, 8) + 1):
            acc = u64(acc + u32(a[i]) * u32(b[k - i]))
        for i in (range(0, k) if k < 9 else range(k - 8, 9)):
            acc = u64(acc + m[i] * p[k - i])
        if k < 9:
            m[k] = ((acc & 0xffffffff) * PINV) & M29
            acc = u64(acc + m[k] * p[0])
            assert acc & M29 == 0
        else:
            r[k - 9] = acc & M29
        acc >>= 29
    r[8] = u32(acc)
    assert value(r) % P == value(a) * value(b) * pow(R261, -1, P) % P
    return r


PTOP1 = (P >> 232) + 1                      # divisor of the quotient estimate
QMAGIC = -(-(1 << 51) // PTOP1)             # ceil(2^51 / d): exact floor(v / d) for v < 2^29
CSUB_P = limbs29(R261 - P)                  # 2^261 - p


def cond_sub_p(x):
    """Field29::cond_sub<0>: x >= p ? x - p : x for a normalized x < 2^261"""
    t = normalize([u32(a + b) for a, b in zip(x, CSUB_P)])
    ge = (t[8] >> 29) != 0
    t[8] &= M29
    return t if ge else list(x)


def canonical(x):
    """last pass without a scaling product: normalized x < 2^261 -> [0, p).  q = floor(top limb / (floor(p / 2^232) + 1)) never exceeds
    floor(x / p) and is at most one short, so x - q p < 2 p; x + q (2^261 - p) mod 2^261 is that difference"""
    assert all(a <= M29 for a in x[:8]) and x[8] <= M29
    q = (x[8] * QMAGIC) >> 51
    assert q == x[8] // PTOP1 and QMAGIC < (1 << 32)
    acc, r = 0, [0] * 9
    for i in range(9):
        acc = u64(acc + q * CSUB_P[i] + x[i])
        r[i] = acc & M29
        acc >>= 29
    assert value(r) == value(x) - q * P and value(r) < 2 * P
    r = cond_sub_p(r)
    assert value(r) == value(x) % P
    return r


def stage_table(log_r, w_r, twist=1):
    """stage-major twiddles of an R-point DIT transform: entry 2^(s-1) - 1 + o = twist^(R / 2^s) w_(2^s)^o in the 2^261 domain"""
    R = 1 << log_r
    t = [0] * R
    for s in range(1, log_r + 1):
        w_s = pow(w_r, R >> s, P)
        tw = pow(twist, R >> s, P)
        for o in range(1 << (s - 1)):
            t[(1 << (s - 1)) - 1 + o] = limbs29(tw * pow(w_s, o, P) * R261 % P)
    return t


def brev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


LDS_STAGES = 6
GROUP = 2


def group_stages(s, log_r):
    """stages the register group starting at stage s covers.  The kernel uses GROUP = 2 (36 data VGPRs); 3 is checked here as well: the limbs
    stay inside 32 bits for three stages between carry passes, it is the register file that says no (250 VGPRs)"""
    left = log_r - s + 1
    if s <= min(log_r, LDS_STAGES):
        left = min(left, min(log_r, LDS_STAGES) - s + 1)
        return min(GROUP, left)
    return min(2, left)


LOOSE_LAST = False          # round 5, measured level and not in the kernel (profiles/r05an_ntt_ab.log): a non-last pass leaving the limbs of its last
                            # register group as they are (the inter-pass product takes loose limbs); the bounds hold (set True to replay them)


def dit_column(col, log_r, table, unit1, stats=None, loose_last=False):
    """the R-point column transform of one tile column as ntt_superstage29 runs it: rows loaded in bit-reversed order, register groups of up
    to three stages with ONE carry propagation at the end, natural order out.  col: R numbers < 2^256 (the data's Montgomery domain)."""
    R = 1 << log_r
    rows = [None] * R
    for i1 in range(R):
        rows[brev(i1, log_r)] = unpack(col[i1])
    s = 1
    while s <= log_r:
        g = group_stages(s, log_r)
        M, h = 1 << g, 1 << (s - 1)
        for q in range(R >> g):
            o, blk = q & (h - 1), q >> (s - 1)
            r0 = (blk << (s - 1 + g)) + o
            x = [rows[r0 + i * h] for i in range(M)]
            for t in range(g):
                half, st = 1 << t, s + t
                base = (1 << (st - 1)) - 1
                for i in range(M):
                    if i & half:
                        continue
                    off = o + (i & (half - 1)) * h
                    if unit1 and st == 1:
                        v = x[i + half]
                        lo = sub(x[i], v, 8)
                    elif SKIP_UNIT2 and unit1 and st == 2 and (i & 1) == 0:
                        # w_4^0 = 1: the partner came out of stage 1 (below 2^256 + 15p, loose limbs) -- one carry pass instead of a product,
                        # and the subtraction borrows 32p
                        assert s == 1 and off == 0
                        v = normalize(x[i + half])
                        lo = sub(x[i], v, 32)
                    else:
                        assert max(x[i + half]) < 6.1 * (1 << 29)
                        v = mont_mul(x[i + half], table[base + off])
                        assert value(v) < 3 * P
                        lo = sub(x[i], v, 4)
                    x[i], x[i + half] = add(x[i], v), lo
            for i in range(M):
                if not (loose_last and LOOSE_LAST and s + g - 1 == log_r):
                    x[i] = normalize(x[i])
                else:
                    assert max(x[i]) < 6.1 * (1 << 29)            # what the inter-pass product accepts
                if stats is not None:
                    stats["max_value_over_p"] = max(stats.get("max_value_over_p", 0), value(x[i]) / P)
                rows[r0 + i * h] = x[i]
        s += g
    return rows


# ---- a whole transform as the host code plans it and the kernel indexes it (ntt.hip: plan_radices, ntt_run_chunk, coset_cm_chunk, ntt_pass_kernel)
NTT_LOG_TILE, NTT_LOG_SINGLE = 10, 11


def plan_radices(log_n, maxr=8):
    if log_n <= NTT_LOG_SINGLE:
        return [log_n]
    np_ = max(2, min(4, (log_n + maxr - 1) // maxr))
    base, extra = divmod(log_n, np_)
    return [base + (1 if i < extra else 0) for i in range(np_)]


def inter_table(log_n, log_m, log_r, w_n, twist=1, log_e=0, b=0, w_ext=None):
    """the inter-pass twiddles of a pass (ntt_twiddle_kernel mode 1; ntt_coset_table_kernel mode 1 for the twisted first pass): entry
    k1 S + i2 = w_M^(i2 k1) [* c_b^i2 for coset b], in the 2^261 domain"""
    S = 1 << (log_m - log_r)
    out = []
    for idx in range(1 << log_m):
        i2, k1 = idx & (S - 1), idx >> (log_m - log_r)
        w = pow(w_n, (i2 * k1 << (log_n - log_m)) % (1 << log_n), P)
        if twist != 1:
            w = w * pow(twist, i2, P) % P
        out.append(limbs29(w * R261 % P))
    return out


def run_pass(src, n_out, log_n, radices, i, table, tw_inter, unit1, in_log_len=None, post=None, coset_pre=None):
    """one launch of ntt_pass_kernel over one column: every tile of the grid, the kernel's own index arithmetic.  post: None, one constant, or
    three (output i is multiplied by post[i mod 3]); coset_pre: (zeta, zeta^2) -- input i of the first pass is multiplied by zeta^(i mod 3)"""
    log_r, npass = radices[i], len(radices)
    first, last = i == 0, i + 1 == len(radices)
    log_m = log_n - sum(radices[:i])
    log_tile = log_n if npass == 1 else max(log_r + 2, NTT_LOG_TILE)
    log_tile = max(log_tile, log_r)
    logC = log_tile - log_r
    C, R, TILE = 1 << logC, 1 << log_r, 1 << log_tile
    log_s = log_m - log_r
    S = 1 << log_s
    in_len = 1 << (log_n if in_log_len is None else in_log_len)
    k1_major = last and npass >= 2 and radices[0] >= logC
    n_blocks = 1 << (log_n - log_r)
    sblk = n_blocks >> radices[0] if npass >= 2 else 1
    out = [None] * n_out
    for tile in range(1 << (log_n - log_tile)):
        def col_base(c):
            if not last:
                colid = tile * C + c
                return ((colid >> log_s) << log_m) + (colid & (S - 1))
            if k1_major:
                rest, k10 = tile % sblk, (tile // sblk) * C
                blk = (k10 + c) * sblk + rest
            else:
                blk = tile * C + c
            return blk << log_r
        for c in range(C):
            col = []
            for i1 in range(R):
                addr = col_base(c) + (i1 << log_s)
                v = src[addr] if (not first or addr < in_len) else 0
                if first and coset_pre is not None and addr < in_len and addr % 3:
                    v = value(mont_mul(unpack(v), limbs29(coset_pre[addr % 3 - 1] * R261 % P)))
                col.append(v)
            rows = dit_column(col, log_r, table, unit1, loose_last=not last)
            for k1 in range(R):
                x = rows[k1]
                if not last:
                    colid = tile * C + c
                    pos = (k1 << log_s) + (colid & (S - 1))
                    out[((colid >> log_s) << log_m) + pos] = pack(mont_mul(x, tw_inter[pos]))
                else:
                    blk = col_base(c) >> log_r
                    oidx, rem, lw, shift = 0, blk, log_n - log_r, 0
                    for p_ in range(npass - 1):
                        lw -= radices[p_]
                        kp = rem >> lw
                        rem &= (1 << lw) - 1
                        oidx += kp << shift
                        shift += radices[p_]
                    oidx += k1 << shift
                    if post is not None:
                        pc = post[oidx % 3] if isinstance(post, (list, tuple)) else post
                        y = cond_sub_p(mont_mul(x, limbs29(pc * R261 % P)))
                    else:
                        y = canonical(x)
                    out[oidx] = pack(y)
    assert all(v is not None for v in out)
    return out


def transform(a, log_n, w_n, inverse_scale=False, in_log_len=None, coset_mode=0, zeta=None):
    """ntt_run_chunk: natural order in and out.  coset_mode 1 = coeff_to_extended in natural order (input i times zeta^(i mod 3), zero-padded
    from 2^in_log_len); 2 = extended_to_coeff (w_n is the inverse root; output i times zeta^-(i mod 3) / n)"""
    radices = plan_radices(log_n)
    table = stage_table(max(radices), pow(w_n, 1 << (log_n - max(radices)), P))
    cur, log_m = list(a), log_n
    post = pow(1 << log_n, -1, P) if (inverse_scale or coset_mode == 2) else None
    if coset_mode == 2:
        z2 = zeta * zeta % P                                   # zeta^-1 = zeta^2, zeta^-2 = zeta
        post = [post, post * z2 % P, post * zeta % P]
    pre = (zeta, zeta * zeta % P) if coset_mode == 1 else None
    for i, lr in enumerate(radices):
        tw = inter_table(log_n, log_m, lr, w_n) if i + 1 < len(radices) else None
        cur = run_pass(cur, 1 << log_n, log_n, radices, i, table[:1 << lr], tw, True, in_log_len if i == 0 else None, post, pre if i == 0 else None)
        log_m -= lr
    return cur


def coset_transform(a, log_n, log_e, w_ext, zeta):
    """coset_cm_chunk: coefficients -> E = 2^log_e coset evaluations, coset b at out[b]: p(zeta w_ext^b w_n^j)"""
    w_n = pow(w_ext, 1 << log_e, P)
    radices = plan_radices(log_n)
    plain = stage_table(max(radices), pow(w_n, 1 << (log_n - max(radices)), P))
    outs = []
    for b in range(1 << log_e):
        c_b = zeta * pow(w_ext, b, P) % P
        cur, log_m = list(a), log_n
        for i, lr in enumerate(radices):
            if i == 0:                                   # the twisted first pass: d = c_b^S on the rows, c_b^i2 in the inter-pass table
                S = 1 << (log_n - lr)
                table = stage_table(lr, pow(w_n, S, P), pow(c_b, S, P))
                tw = inter_table(log_n, log_m, lr, w_n, twist=c_b) if len(radices) > 1 else None
                cur = run_pass(cur, 1 << log_n, log_n, radices, i, table, tw, False)
            else:
                tw = inter_table(log_n, log_m, lr, w_n) if i + 1 < len(radices) else None
                cur = run_pass(cur, 1 << log_n, log_n, radices, i, plain[:1 << lr], tw, True)
            log_m -= lr
        outs.append(cur)
    return outs


def selftest(seed=1, log_rs=(1, 2, 3, 5, 6, 8, 11), group=2):
    global GROUP
    old_group, GROUP = GROUP, group
    try:
        return _selftest(seed, log_rs)
    finally:
        GROUP = old_group                       # the kernel's group size for everything that runs after


def _selftest(seed, log_rs):
    rnd = random.Random(seed)
    gen = pow(7, (P - 1) >> 28, P)
    worst = {}
    for log_r in log_rs:
        R = 1 << log_r
        w_r = pow(gen, 1 << (28 - log_r), P)
        for twist in (1, pow(5, 12345, P)):
            table = stage_table(log_r, w_r, twist)
            # the loader accepts anything below 2^256 (lazily reduced work buffers, non-canonical callers)
            col = [rnd.randrange(1 << 256) if rnd.random() < 0.5 else rnd.randrange(P) for _ in range(R)]
            if log_r >= 3:
                col[0], col[1], col[2] = (1 << 256) - 1, 0, P - 1
            rows = dit_column(col, log_r, table, unit1=(twist == 1), stats=worst)
            tpow = [pow(twist, i, P) for i in range(R)]
            ks = range(R) if R <= 64 else rnd.sample(range(R), 24)
            for k in ks:
                want = sum(col[i] * tpow[i] * pow(w_r, i * k, P) for i in range(R)) % P
                got = canonical(rows[k])
                assert value(got) == want, (log_r, twist != 1, k)
                # non-last passes: the inter-pass product brings the value back below 2^256 for the work buffer
                tw = limbs29(rnd.randrange(P))
                y = mont_mul(rows[k], tw)
                assert pack(y) < 2 * P
    # conversions and the quotient estimate at the edges
    for x in (0, 1, P - 1, P, 2 * P - 1, (1 << 256) - 1, R261 - 1, 168 * P + 5):
        assert value(canonical(normalize(limbs29(x)))) == x % P
    for _ in range(2000):
        x = rnd.randrange(R261)
        assert value(canonical(limbs29(x))) == x % P
        w = rnd.randrange(1 << 256)
        assert pack(unpack(w)) == w
    return worst


if __name__ == "__main__":
    print(selftest())
