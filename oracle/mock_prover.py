"""MockProver for ezkl_amd constraint systems (TEST INFRASTRUCTURE ONLY): the role halo2_proofs::dev::MockProver plays in the
reference's circuit tests (/root/reference/src/circuit/tests.rs:26-100 ...): checks an assignment row by row with plain
Python big-ints -- every gate polynomial on every usable row, every lookup input tuple against the set of table rows,
every copy constraint -- and reports the failures instead of producing a proof."""
from ezkl_amd import plonk as P

R = P.R


def _ev(e, q, memo):
    k = id(e)
    if k in memo:
        return memo[k]
    op = e.node[0]
    if op == "const": r = e.node[1]
    elif op == "chal": r = q("chal", e.node[1], 0)
    elif op in ("adv", "fix", "inst"): r = q(op, e.node[1], e.node[2])
    elif op == "neg": r = (-_ev(e.node[1], q, memo)) % R
    else:
        a, b = _ev(e.node[1], q, memo), _ev(e.node[2], q, memo)
        r = (a + b) % R if op == "add" else (a - b) % R if op == "sub" else a * b % R
    memo[k] = r
    return r


def check(cs, advice, fixed, instance=(), copies=(), challenges=(), max_failures=8):
    """cs: plonk.ConstraintSystem; advice / fixed / instance: lists of columns, each a list of n ints (instance columns may be
    shorter: zero-padded); copies: [((colpos, row), (colpos, row))] over cs.perm.  Returns a list of failure strings."""
    n, u = cs.n, cs.usable
    inst = [list(c) + [0] * (n - len(c)) for c in instance]
    cols = {"adv": advice, "fix": fixed, "inst": inst}
    fails = []

    def at(row):
        def q(kind, c, rot):
            if kind == "chal":
                return challenges[c]
            return int(cols[kind][c][(row + rot) % n])
        return q
    for row in range(u):
        memo, q = {}, at(row)
        for gi, g in enumerate(cs.gates):
            if _ev(g, q, memo) != 0:
                fails.append("gate %d not satisfied on row %d" % (gi, row))
                if len(fails) >= max_failures: return fails
    for li, (ins, tab) in enumerate(cs.lookups):
        rows = set()
        for row in range(u):
            memo, q = {}, at(row)
            rows.add(tuple(_ev(e, q, memo) for e in tab))
        for ti, t in enumerate(ins):
            for row in range(u):
                memo, q = {}, at(row)
                v = tuple(_ev(e, q, memo) for e in t)
                if v not in rows:
                    fails.append("lookup %d input %d row %d: %s not in table" % (li, ti, row, [x if x < R // 2 else x - R for x in v]))
                    if len(fails) >= max_failures: return fails
    for (ca, ra), (cb, rb) in copies:
        ka, ia = cs.perm[ca]
        kb, ib = cs.perm[cb]
        if cols[ka][ia][ra] != cols[kb][ib][rb]:
            fails.append("copy (%s%d,%d) != (%s%d,%d)" % (ka, ia, ra, kb, ib, rb))
            if len(fails) >= max_failures: return fails
    return fails
