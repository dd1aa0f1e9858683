"""Independent verifier for ezkl_amd.plonk proofs (TEST INFRASTRUCTURE ONLY).

Plain Python big-ints + the oracle pairing.  It shares NO code with the prover except the transcript hash and the
constraint-system description: expressions are re-evaluated from the claimed evaluations, the quotient identity is
checked at x, and the SHPLONK opening is checked with a real BN254 pairing against the G2 elements of the SRS.  This is
the acceptance criterion the reference uses for its own proofs (verify_proof_circuit, src/pfsys/mod.rs:557-590)."""
from ezkl_amd.transcript import EvmTranscript, R
from ezkl_amd import plonk as P
from . import pairing as E


def _lagrange_evals(k, x, rows):
    n = 1 << k
    w = P.omega(k)
    zx = (pow(x, n, R) - 1) % R
    ninv = pow(n, -1, R)
    return {i: pow(w, i, R) * zx % R * ninv % R * pow((x - pow(w, i, R)) % R, -1, R) % R for i in rows}


def verify(vk, g1_gen, g2, s_g2, proof, instances=(), dbg=None):
    """dbg: a dict that receives the intermediate values (challenges, Lagrange evaluations, quotient numerator, pairing inputs)"""
    cs = vk.cs
    n, k, u = cs.n, cs.k, cs.usable
    try:
        T = EvmTranscript(proof)
        T.common_scalar(vk.digest)
        if len(instances) != cs.n_instance:
            return False
        for vals in instances:
            for v_ in vals:
                T.common_scalar(v_)
        adv_c = [None] * cs.n_advice
        user_chal = []
        for phase in (0, 1):
            idxs = [c for c in range(cs.n_advice) if cs.advice_phase[c] == phase]
            for c in idxs:
                adv_c[c] = T.read_point()
            if phase == 0 and idxs:
                user_chal = [T.squeeze_challenge() for _ in range(cs.n_challenges)]
        theta, m_c = None, []
        if cs.lookups:
            theta = T.squeeze_challenge()
            m_c = [T.read_point() for _ in cs.lookups]
        beta, gamma = T.squeeze_challenge(), T.squeeze_challenge()
        z_c = [T.read_point() for _ in range(cs.n_chunks)]
        phi_c = [T.read_point() for _ in cs.lookups]
        rnd_c = T.read_point()
        y = T.squeeze_challenge()
        h_c = [T.read_point() for _ in range(cs.degree - 1)]
        x = T.squeeze_challenge()
        ev = {}
        for c, r in cs.advice_queries: ev[("adv", c, r)] = T.read_scalar()
        for c, r in cs.fixed_queries: ev[("fix", c, r)] = T.read_scalar()
        random_eval = T.read_scalar()
        sigma_ev = [T.read_scalar() for _ in cs.perm]
        z_ev = []
        for j in range(cs.n_chunks):
            e0, e1 = T.read_scalar(), T.read_scalar()
            e2 = T.read_scalar() if j + 1 < cs.n_chunks else None
            z_ev.append((e0, e1, e2))
        lk_ev = [(T.read_scalar(), T.read_scalar(), T.read_scalar()) for _ in cs.lookups]
    except ValueError:
        return False
    w = P.omega(k)
    def rot_point(r): return x * pow(w, r % n, R) % R
    # ---- expected quotient evaluation
    # instance evaluations from the public values: inst(z) = sum_i v_i * l_i(z)
    zx_cache = {}
    def inst_eval(c, r):
        z = rot_point(r)
        vals = instances[c]
        if not vals:
            return 0
        le = _lagrange_evals(k, z, range(len(vals)))
        return sum(v_ * le[i] for i, v_ in enumerate(vals)) % R
    for c, r in cs.instance_queries:
        ev[("inst", c, r)] = inst_eval(c, r)
    for i_, cval in enumerate(user_chal):
        ev[("chal", i_, 0)] = cval
    lag = _lagrange_evals(k, x, [0] + list(range(u, n)))
    l0, llast = lag[0], lag[u]
    lact = (1 - sum(lag[i] for i in range(u, n))) % R
    terms = [P.evaluate(g, lambda kd, c, r: ev[(kd, c, r)]) for g in cs.gates]
    if cs.perm:
        terms.append(l0 * (1 - z_ev[0][0]) % R)
        zl = z_ev[-1][0]
        terms.append(llast * (zl * zl - zl) % R)
        for j in range(1, cs.n_chunks):
            terms.append(l0 * (z_ev[j][0] - z_ev[j - 1][2]) % R)
        pos = 0
        for j, chunk in enumerate(cs.perm_chunks()):
            left, right = z_ev[j][1], z_ev[j][0]
            for i, (kd, c) in enumerate(chunk):
                v = ev[(kd, c, 0)]
                left = left * (v + beta * sigma_ev[pos + i] + gamma) % R
                right = right * (v + beta * pow(P.DELTA, pos + i, R) % R * x + gamma) % R
            terms.append(lact * (left - right) % R)
            pos += len(chunk)
    q = lambda kd, c, r: ev[(kd, c, r)]
    def compress(tup):
        acc = P.evaluate(tup[0], q)
        for e in tup[1:]:
            acc = (acc * theta + P.evaluate(e, q)) % R
        return acc
    for (ins, tab), (phi_e, phi_n, m_e) in zip(cs.lookups, lk_ev):
        fb = [(compress(t) + beta) % R for t in ins]
        tb = (compress(tab) + beta) % R
        prodf = 1
        for f in fb: prodf = prodf * f % R
        ssum = 0
        for j in range(len(fb)):
            pj = 1
            for i2, f in enumerate(fb):
                if i2 != j: pj = pj * f % R
            ssum = (ssum + pj) % R
        lhs = (phi_n - phi_e) * prodf % R * tb % R
        rhs = (ssum * tb - m_e * prodf) % R
        terms += [l0 * phi_e % R, llast * phi_e % R, lact * (lhs - rhs) % R]
    num = 0
    for t in terms:
        num = (num * y + t) % R
    xn = pow(x, n, R)
    h_eval = num * pow((xn - 1) % R, -1, R) % R
    if dbg is not None:
        dbg.update(theta=theta, beta=beta, gamma=gamma, y=y, x=x, l0=l0, llast=llast, lact=lact, terms=list(terms), num=num, h_eval=h_eval, xn=xn)
    # ---- queries, in the prover's order
    hc = None
    for c in reversed(h_c):
        hc = E.g1_add(E.g1_mul(hc, xn), c)
    qs = []
    for c, r in cs.advice_queries: qs.append((("adv", c), adv_c[c], rot_point(r), ev[("adv", c, r)]))
    for j in range(cs.n_chunks):
        qs.append((("z", j), z_c[j], x, z_ev[j][0])); qs.append((("z", j), z_c[j], rot_point(1), z_ev[j][1]))
        if z_ev[j][2] is not None: qs.append((("z", j), z_c[j], rot_point(u), z_ev[j][2]))
    for i, (phi_e, phi_n, m_e) in enumerate(lk_ev):
        qs.append((("phi", i), phi_c[i], x, phi_e)); qs.append((("phi", i), phi_c[i], rot_point(1), phi_n))
        qs.append((("m", i), m_c[i], x, m_e))
    for c, r in cs.fixed_queries: qs.append((("fix", c), vk.fixed_commitments[c], rot_point(r), ev[("fix", c, r)]))
    for i, e in enumerate(sigma_ev): qs.append((("sigma", i), vk.sigma_commitments[i], x, e))
    qs.append((("h",), hc, x, h_eval))
    qs.append((("rnd",), rnd_c, x, random_eval))
    # ---- SHPLONK
    groups = P.group_queries(qs)
    ys = T.squeeze_challenge()
    all_pts = sorted({z for pts, _ in groups for z in pts})
    v = T.squeeze_challenge()
    try:
        pi1 = T.read_point()
        uu = T.squeeze_challenge()
        pi2 = T.read_point()
    except ValueError:
        return False
    if T._rd != len(proof):
        return False
    L, pw, norm, z0 = None, 1, None, 1
    for gi, (pts, polys) in enumerate(groups):
        qc, evs, yp = None, {z: 0 for z in pts}, 1
        for c, e in polys:
            qc = E.g1_add(qc, E.g1_mul(c, yp))
            for z in pts: evs[z] = (evs[z] + yp * e[z]) % R
            yp = yp * ys % R
        r = P.interpolate(list(pts), [evs[z] for z in pts])
        zdiff = 1
        for z in all_pts:
            if z not in pts: zdiff = zdiff * (uu - z) % R
        if gi == 0:                                     # halo2: coefficients normalised by the first set's; -Z_{S_0}(u) on the first opening point
            norm = pow(zdiff, -1, R)
            for z in pts: z0 = z0 * (uu - z) % R
        term = E.g1_add(qc, E.g1_neg(E.g1_mul(g1_gen, P.eval_small(r, uu))))
        L = E.g1_add(L, E.g1_mul(term, pw * zdiff % R * norm % R))
        pw = pw * v % R
    L = E.g1_add(L, E.g1_neg(E.g1_mul(pi1, z0)))
    lhs = E.g1_add(L, E.g1_mul(pi2, uu))
    if dbg is not None:
        dbg.update(shplonk_y=ys, shplonk_v=v, shplonk_u=uu, pairing_lhs=lhs, pairing_rhs=pi2)
    return E.pairing_check([(pi2, s_g2), (E.g1_neg(lhs), g2)])
