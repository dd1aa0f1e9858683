"""End-to-end: the halo2-shaped prover (ezkl_amd/plonk.py) makes a proof that the independent pairing verifier
accepts (the reference's own acceptance criterion), on the reference's k=6 SRS fixture.
CPU: protocol logic on the oracle backend.  GPU: the same proof BYTES from the HIP kernels."""
import numpy as np
import pytest
from conftest import fe_from_int, R, json_lines
from ezkl_amd import plonk as P
from oracle import pairing as E, pyref as pr, verifier as V


def mul_add_circuit(k):
    """a small ezkl-flavoured circuit: per row  sel_mul*(c - a*b) = 0  and  sel_acc*(c - c[-1] - a*b) = 0
    (MULT and DOT-style gates, src/circuit/ops/chip.rs:344-425), with copy constraints chaining rows"""
    a, b, c = P.adv(0), P.adv(1), P.adv(2)
    sel_mul, sel_acc = P.fix(0), P.fix(1)
    gates = [sel_mul * (c - a * b), sel_acc * (c - P.adv(2, -1) - a * b)]
    perm = [("adv", 0), ("adv", 1), ("adv", 2), ("fix", 2)]
    return P.ConstraintSystem(k, 3, 3, gates, perm)


def witness(cs, seed):
    rng = np.random.default_rng(seed)
    n, u = cs.n, cs.usable
    A = [[0] * n for _ in range(3)]
    F = [[0] * n for _ in range(3)]
    copies = []
    half = u // 2
    for r in range(half):                      # MULT rows
        x, y = int(rng.integers(1, 1 << 30)), int(rng.integers(1, 1 << 30))
        A[0][r], A[1][r], A[2][r] = x, y, x * y % R
        F[0][r] = 1
    A[2][half] = 0                              # accumulator start (c[half] is a free cell constrained via copy to constant 0)
    F[2][half] = 0
    copies.append(((2, half), (3, half)))       # adv2[half] == fix2[half] (= 0)
    for r in range(half + 1, u):               # DOT rows: c = c[-1] + a*b, inputs copied from the MULT rows
        src = r - half - 1
        A[0][r], A[1][r] = A[0][src], A[1][src]
        copies.append(((0, r), (0, src)))
        copies.append(((1, r), (1, src)))
        A[2][r] = (A[2][r - 1] + A[0][r] * A[1][r]) % R
        F[1][r] = 1
    to_col = lambda col: np.stack([fe_from_int(v) for v in col])
    return [to_col(c) for c in A], [to_col(c) for c in F], copies


def lookup_circuit(k):
    """matmul-style gate + a two-column lookup (x, relu(x - 8)) in (table_in, table_out): the shape of ezkl's
    nonlinearity lookups (src/circuit/ops/chip.rs:484-599) with a theta-compressed 2-tuple, plus a second
    single-column input set sharing a range table"""
    a, b, c, o = P.adv(0), P.adv(1), P.adv(2), P.adv(3)
    sel = P.fix(0)
    gates = [sel * (c - a * b)]
    lookups = [([[sel * a, sel * o]], [P.fix(1), P.fix(2)]),            # (a, o) in {(x, relu(x-8))} on selected rows, (0,0) elsewhere
               ([[sel * b], [sel * a]], [P.fix(1)])]                    # b and a in the range table: two input sets, one table
    perm = [("adv", 0), ("adv", 3)]
    return P.ConstraintSystem(k, 4, 3, gates, perm, lookups)


def lookup_witness(cs, seed, bad=None):
    rng = np.random.default_rng(seed)
    n, u = cs.n, cs.usable
    A = [[0] * n for _ in range(4)]
    F = [[0] * n for _ in range(3)]
    T = 16
    for r in range(u):                                 # table rows: x = r mod 16, relu(x - 8); row (0, 0) included
        x = r % T
        F[1][r], F[2][r] = x, max(x - 8, 0)
    for r in range(u // 2):
        x, y = int(rng.integers(0, T)), int(rng.integers(0, T))
        A[0][r], A[1][r], A[2][r], A[3][r] = x, y, x * y, max(x - 8, 0)
        F[0][r] = 1
    copies = [((0, 1), (0, 0))] if A[0][1] == A[0][0] else []
    if bad == "lookup":
        A[3][2] = (A[3][2] + 1) % R                    # (a, o) no longer in the table
    if bad == "range":
        A[1][4] = 99                                   # b out of range (and c adjusted so the gate still holds)
        A[2][4] = A[0][4] * 99
    to_col = lambda col: np.stack([fe_from_int(v) for v in col])
    return [to_col(c) for c in A], [to_col(c) for c in F], copies


def test_lookup_argument_oracle_backend(golden_srs):
    from oracle.cpu_backend import OracleBackend
    cs = lookup_circuit(6)
    assert cs.degree == 7 and cs.ext_k == 9      # l_active * phi * (sel*b + beta)(sel*a + beta) * (t + beta)
    be = OracleBackend(golden_srs["g"], golden_srs["g_lagrange"], 6)
    g1, g2, s_g2 = setup(golden_srs)
    adv, fixed, copies = lookup_witness(cs, 3)
    pk, vk = P.keygen(cs, be, fixed, copies)
    assert V.verify(vk, g1, g2, s_g2, P.create_proof(pk, be, adv, det_rng(1)))
    for bad in ("lookup", "range"):
        adv_b, _, _ = lookup_witness(cs, 3, bad=bad)
        with pytest.raises(ValueError, match="not in table"):          # the prover refuses, as the reference's mv-lookup prover does
            P.create_proof(pk, be, adv_b, det_rng(1))
        assert not V.verify(vk, g1, g2, s_g2, P.create_proof(pk, be, adv_b, det_rng(1), strict=False))


@pytest.mark.gpu
def test_gpu_lookup_proof_bit_identical(hip, golden_srs):
    from oracle.cpu_backend import OracleBackend
    cs = lookup_circuit(6)
    adv, fixed, copies = lookup_witness(cs, 4)
    cpu = OracleBackend(golden_srs["g"], golden_srs["g_lagrange"], 6)
    gpu = P.GpuBackend(golden_srs["g"], golden_srs["g_lagrange"], 6)
    pk_c, vk_c = P.keygen(cs, cpu, fixed, copies)
    pk_g, vk_g = P.keygen(cs, gpu, fixed, copies)
    proof_c = P.create_proof(pk_c, cpu, adv, det_rng(2))
    proof_g = P.create_proof(pk_g, gpu, adv, det_rng(2))
    assert proof_g == proof_c
    g1, g2, s_g2 = setup(golden_srs)
    assert V.verify(vk_g, g1, g2, s_g2, proof_g)


def instance_phase_circuit(k):
    """public output through an instance column (as ezkl exposes model outputs, src/graph/mod.rs:1411-1446) and a
    second-phase advice column constrained with a post-commitment challenge (the Freivalds / RLC pattern of
    src/circuit/ops/chip/einsum/mod.rs:715-783): c = a + r*b with r squeezed after a, b are committed"""
    a, b, c, acc = P.adv(0), P.adv(1), P.adv(2), P.adv(3)
    sel = P.fix(0)
    gates = [sel * (c - a - P.chal(0) * b),                       # phase-1 column against the challenge
             sel * (acc - P.adv(3, -1) - a * b) + (1 - sel) * 0]   # running dot product on selected rows (row 0 starts from acc[-1])
    perm = [("adv", 3), ("inst", 0), ("adv", 0)]
    return P.ConstraintSystem(k, 4, 1, gates, perm, n_instance=1, advice_phase=[0, 0, 1, 0], n_challenges=1)


def instance_phase_witness(cs, seed):
    rng = np.random.default_rng(seed)
    n, u = cs.n, cs.usable
    a = [int(rng.integers(1, 1 << 20)) for _ in range(n)]
    b = [int(rng.integers(1, 1 << 20)) for _ in range(n)]
    acc = [0] * n
    sel = [0] * n
    for r in range(1, u):
        sel[r] = 1
        acc[r] = (acc[r - 1] + a[r] * b[r]) % R
    public = acc[u - 1]
    to_col = lambda col: np.stack([fe_from_int(v) for v in col])
    cols0 = {0: to_col(a), 1: to_col(b), 3: to_col(acc)}
    def advice(phase, challenges):
        if phase == 0:
            return cols0
        r_ = challenges[0]
        return {2: to_col([(a[i] + r_ * b[i]) % R for i in range(n)])}
    copies = [((0, u - 1), (1, 0))]                      # adv3[u-1] == inst0[0]
    return advice, [to_col(sel)], copies, [[public]]


def test_instance_and_second_phase_oracle_backend(golden_srs):
    from oracle.cpu_backend import OracleBackend
    cs = instance_phase_circuit(6)
    be = OracleBackend(golden_srs["g"], golden_srs["g_lagrange"], 6)
    g1, g2, s_g2 = setup(golden_srs)
    advice, fixed, copies, inst = instance_phase_witness(cs, 5)
    pk, vk = P.keygen(cs, be, fixed, copies)
    proof = P.create_proof(pk, be, advice, det_rng(3), instances=inst)
    assert V.verify(vk, g1, g2, s_g2, proof, instances=inst)
    assert not V.verify(vk, g1, g2, s_g2, proof, instances=[[(inst[0][0] + 1) % R]])      # wrong public output
    assert not V.verify(vk, g1, g2, s_g2, proof)                                            # missing instances


@pytest.mark.gpu
def test_gpu_instance_second_phase_bit_identical(hip, golden_srs):
    from oracle.cpu_backend import OracleBackend
    cs = instance_phase_circuit(6)
    advice, fixed, copies, inst = instance_phase_witness(cs, 6)
    cpu = OracleBackend(golden_srs["g"], golden_srs["g_lagrange"], 6)
    gpu = P.GpuBackend(golden_srs["g"], golden_srs["g_lagrange"], 6)
    pk_c, _ = P.keygen(cs, cpu, fixed, copies)
    pk_g, vk_g = P.keygen(cs, gpu, fixed, copies)
    proof_c = P.create_proof(pk_c, cpu, advice, det_rng(4), instances=inst)
    proof_g = P.create_proof(pk_g, gpu, advice, det_rng(4), instances=inst)
    assert proof_g == proof_c
    g1, g2, s_g2 = setup(golden_srs)
    assert V.verify(vk_g, g1, g2, s_g2, proof_g, instances=inst)


def random_circuit(seed, k=6):
    """random gate set  sel_g * (out_g - f_g(inputs))  with f_g a random polynomial of degree <= 3 over rotated advice
    cells (always satisfiable: out is computed from f), random copy constraints between equal cells, a random
    single-column range lookup"""
    rng = np.random.default_rng(seed)
    n_in, n_gates = 3, int(rng.integers(1, 4))
    n = 1 << k
    u = n - P.BLINDING - 1

    def rand_poly():
        terms = []
        for _ in range(int(rng.integers(1, 4))):
            deg = int(rng.integers(1, 4))
            factors = [(int(rng.integers(0, n_in)), int(rng.integers(-1, 2))) for _ in range(deg)]
            terms.append((int(rng.integers(1, 1000)), factors))
        return terms

    polys = [rand_poly() for _ in range(n_gates)]
    gates = []
    for g, terms in enumerate(polys):
        e = P.const(0)
        for coef, factors in terms:
            t = P.const(coef)
            for c, r in factors:
                t = t * P.adv(c, r)
            e = e + t
        gates.append(P.fix(g) * (P.adv(n_in + g) - e))
    lookups = [([[P.fix(n_gates) * P.adv(0)]], [P.fix(n_gates + 1)])]
    perm = [("adv", c) for c in range(n_in)] + [("fix", n_gates + 2)]
    cs = P.ConstraintSystem(k, n_in + n_gates, n_gates + 3, gates, perm, lookups)
    # witness
    A = [[0] * n for _ in range(n_in + n_gates)]
    F = [[0] * n for _ in range(n_gates + 3)]
    T = 32
    for c in range(n_in):
        for r in range(n):
            A[c][r] = int(rng.integers(0, T))
    copies, used = [], set()
    while len(copies) < 6:                                # tie random (distinct) cells of the input columns together
        c1, c2 = int(rng.integers(0, n_in)), int(rng.integers(0, n_in))
        r1, r2 = int(rng.integers(1, u - 1)), int(rng.integers(1, u - 1))
        if (c1, r1) in used or (c2, r2) in used or (c1, r1) == (c2, r2):
            continue
        used.update([(c1, r1), (c2, r2)])
        A[c2][r2] = A[c1][r1]
        copies.append(((c1, r1), (c2, r2)))
    for r in range(u):
        F[n_gates + 1][r] = r % T                         # range table 0..31
    for r in range(1, u - 1):
        F[n_gates][r] = 1                                 # lookup selector
        for g, terms in enumerate(polys):
            F[g][r] = 1
            acc = 0
            for coef, factors in terms:
                t = coef
                for c, rot in factors:
                    t = t * A[c][(r + rot) % n] % R
                acc = (acc + t) % R
            A[n_in + g][r] = acc
    to_col = lambda col: np.stack([fe_from_int(v) for v in col])
    return cs, [to_col(c) for c in A], [to_col(c) for c in F], copies


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_circuits_oracle_backend(golden_srs, seed):
    from oracle.cpu_backend import OracleBackend
    cs, adv, fixed, copies = random_circuit(seed)
    be = OracleBackend(golden_srs["g"], golden_srs["g_lagrange"], 6)
    g1, g2, s_g2 = setup(golden_srs)
    pk, vk = P.keygen(cs, be, fixed, copies)
    proof = P.create_proof(pk, be, adv, det_rng(seed))
    assert V.verify(vk, g1, g2, s_g2, proof)
    bad = [a.copy() for a in adv]
    bad[cs.n_advice - 1][5] = fe_from_int((P.from_mont(bad[cs.n_advice - 1][5]) + 1) % R)     # break the last gate on row 5
    assert not V.verify(vk, g1, g2, s_g2, P.create_proof(pk, be, bad, det_rng(seed)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [21, 22])
def test_random_circuits_gpu_bit_identical(hip, golden_srs, seed):
    from oracle.cpu_backend import OracleBackend
    cs, adv, fixed, copies = random_circuit(seed)
    cpu = OracleBackend(golden_srs["g"], golden_srs["g_lagrange"], 6)
    gpu = P.GpuBackend(golden_srs["g"], golden_srs["g_lagrange"], 6)
    pk_c, _ = P.keygen(cs, cpu, fixed, copies)
    pk_g, vk_g = P.keygen(cs, gpu, fixed, copies)
    pc, pg = P.create_proof(pk_c, cpu, adv, det_rng(seed)), P.create_proof(pk_g, gpu, adv, det_rng(seed))
    assert pc == pg
    g1, g2, s_g2 = setup(golden_srs)
    assert V.verify(vk_g, g1, g2, s_g2, pg)


def det_rng(seed):
    return P.Rng(seed)


def setup(golden_srs):
    srs = pr.parse_srs(golden_srs["buf"])
    g2, s_g2 = E.g2_from_bytes(srs["g2"]), E.g2_from_bytes(srs["s_g2"])
    g1 = pr.g1_from_bytes(srs["g"][0])
    return g1, g2, s_g2


def test_prove_verify_oracle_backend(golden_srs):
    from oracle.cpu_backend import OracleBackend
    cs = mul_add_circuit(6)
    adv, fixed, copies = witness(cs, 1)
    be = OracleBackend(golden_srs["g"], golden_srs["g_lagrange"], 6)
    pk, vk = P.keygen(cs, be, fixed, copies)
    proof = P.create_proof(pk, be, adv, det_rng(7))
    g1, g2, s_g2 = setup(golden_srs)
    assert V.verify(vk, g1, g2, s_g2, proof)
    # tampering: any flipped byte, or a witness that violates a gate / a copy constraint, must be rejected
    bad = bytearray(proof); bad[len(bad) // 2] ^= 1
    assert not V.verify(vk, g1, g2, s_g2, bytes(bad))
    adv_bad = [a.copy() for a in adv]; adv_bad[2][3] = fe_from_int(12345)          # breaks c = a*b on row 3
    assert not V.verify(vk, g1, g2, s_g2, P.create_proof(pk, be, adv_bad, det_rng(7)))
    adv_bad = [a.copy() for a in adv]; adv_bad[0][cs.usable - 1] = adv_bad[0][0]    # breaks a copy constraint, keeps the gate consistent
    r = cs.usable - 1
    adv_bad[2][r] = fe_from_int((P.from_mont(adv_bad[2][r - 1]) + P.from_mont(adv_bad[0][r]) * P.from_mont(adv_bad[1][r])) % R)
    assert not V.verify(vk, g1, g2, s_g2, P.create_proof(pk, be, adv_bad, det_rng(7)))


def test_export_keys_in_halo2_layout(golden_srs):
    """keygen output serialises to the vk.key / pk.key raw-bytes layout and parses back"""
    from oracle.cpu_backend import OracleBackend
    from ezkl_amd import codecs
    cs = mul_add_circuit(6)
    adv, fixed, copies = witness(cs, 1)
    be = OracleBackend(golden_srs["g"], golden_srs["g_lagrange"], 6)
    pk, vk = P.keygen(cs, be, fixed, copies)
    vkb, pkb = P.export_keys(pk, be)
    assert pkb.startswith(vkb) and vkb[0] == 3 and vkb[1] == 6
    back = codecs.read_pk(pkb, n_perm=len(cs.perm), n_selectors=0)
    assert len(back["fixed_polys"]) == cs.n_fixed and back["l0"].shape == (1 << cs.ext_k, 4)
    assert (back["perm_cosets"][1] == be.download(pk.sigma_cosets[1], 1 << cs.ext_k)).all()
    # the permutation polynomials of the identity part are the (0, delta^j, 0, ...) vectors the reference's pk.key shows
    ident = [j for j, col in enumerate(back["perm_polys"]) if not col[2:].any() and not col[0].any()]
    for j in ident:
        assert P.from_mont(back["perm_polys"][j][1]) == pow(P.DELTA, j, R)


@pytest.mark.gpu
def test_gpu_proof_is_bit_identical_to_cpu_proof(hip, golden_srs):
    """north-star: proofs bit-identical to the CPU prover on the same SRS / witness / randomness"""
    from oracle.cpu_backend import OracleBackend
    cs = mul_add_circuit(6)
    adv, fixed, copies = witness(cs, 2)
    cpu = OracleBackend(golden_srs["g"], golden_srs["g_lagrange"], 6)
    gpu = P.GpuBackend(golden_srs["g"], golden_srs["g_lagrange"], 6)
    pk_c, vk_c = P.keygen(cs, cpu, fixed, copies)
    pk_g, vk_g = P.keygen(cs, gpu, fixed, copies)
    assert vk_c.digest == vk_g.digest and vk_c.fixed_commitments == vk_g.fixed_commitments
    proof_c = P.create_proof(pk_c, cpu, adv, det_rng(11))
    proof_g = P.create_proof(pk_g, gpu, adv, det_rng(11))
    assert proof_g == proof_c
    g1, g2, s_g2 = setup(golden_srs)
    assert V.verify(vk_g, g1, g2, s_g2, proof_g)


@pytest.mark.gpu
def test_msm_sharded_prover_two_ranks_same_proof(hip):
    """BASELINE configs[3] plumbing on one GPU: two ranks (gloo, both on GPU 0) each hold half of the SRS, shard every
    MSM of the proof by points, all_gather the partials; the proof equals the single-rank proof and verifies"""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, K="12", BLOCKS="1", CIRCUIT="synthetic")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prove_bench.py")], env=env, capture_output=True, text=True, timeout=600)
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", os.path.join(ROOT, "tools", "prove_bench.py"), "--share-device", "--gloo", "--native"],
                         env=env, capture_output=True, text=True, timeout=900)
    objs = json_lines(two.stdout)
    assert objs, two.stderr[-2000:]
    j2 = objs[-1]
    assert j1["verifier_accepts"] and j2["verifier_accepts"]
    assert j2["n_gpus"] == 2 and j1["proof_sha256"] == j2["proof_sha256"]
    assert "2 sharded sweep(s)" in j2["sweep_sharding"] and "0 sharded" in j1["sweep_sharding"]      # warm-up + timed proof, rows split over the ranks
    # the C++ host prover sharded the same way (ezkl_prover_cs_set_shard): same bytes as the Python host, same on both ranks
    nv = j2["native_prover"]
    assert nv["proof_identical_to_python_prover"] and nv["all_ranks_same_proof"] and nv["library_rng_proof_verifies"]
    assert nv["sharded_sweeps"] == 2                                   # warm-up + timed proof: the sweep split by rows, h all_gathered
    assert nv["commit_sharding"].startswith("by columns")              # default: complete base sets, commit batches divided by columns
    # the same with 1/world of the SRS per rank (every MSM divided by points): same proof again
    two_s = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                            "--master-port", "29543", os.path.join(ROOT, "tools", "prove_bench.py"), "--share-device", "--gloo", "--native", "--slice-bases"],
                           env=env, capture_output=True, text=True, timeout=900)
    objs = json_lines(two_s.stdout)
    assert objs, two_s.stderr[-2000:]
    j3 = objs[-1]
    assert j3["native_prover"]["commit_sharding"].startswith("by points") and j3["native_prover"]["proof_identical_to_python_prover"]
    assert j3["native_prover"]["all_ranks_same_proof"] and j3["native_prover"]["library_rng_proof_verifies"]


@pytest.mark.gpu
def test_column_sharded_ntt_prover_two_ranks_same_proof(hip):
    """SURVEY.md §8(e) in full on the device: MSMs by points, NTTs by COLUMNS (each rank transforms only the columns it owns), the sweep
    by rows fed by the all-to-all of coset windows, evaluations and SHPLONK partial sums by owner -- two ranks (gloo, sharing the one
    GPU of the test box) emit the single-rank proof"""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, K="12", BLOCKS="2", CIRCUIT="synthetic")
    one = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prove_bench.py")], env=env, capture_output=True, text=True, timeout=600)
    j1 = json.loads(one.stdout.strip().splitlines()[-1])
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29547", os.path.join(ROOT, "tools", "prove_bench.py"), "--share-device", "--gloo", "--shard-columns"],
                         env=env, capture_output=True, text=True, timeout=900)
    objs = json_lines(two.stdout)
    assert objs, two.stderr[-2000:]
    j2 = objs[-1]
    assert j1["verifier_accepts"] and j2["verifier_accepts"] and j1["proof_sha256"] == j2["proof_sha256"]
    assert j2["ntt_sharding"].startswith("columns round-robin across 2 ranks") and j1["ntt_sharding"] == "replicated"


@pytest.mark.gpu
def test_bench_contract_two_ranks(hip, tmp_path):
    """`bench.py --gpus 2` the way the driver launches it (torchrun), on one GPU with gloo: ONE JSON line from rank 0 with the
    contract's keys -- short enough for the driver's record -- the whole-job value, and the sharded end-to-end prove leg accepted by the
    verifier (its details in the full record, bench_full.json)"""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, EZKL_BENCH_PROVE_TIMEOUT="200", EZKL_BENCH_MULTI_MLP20="0", EZKL_BENCH_MULTI_K22="11", EZKL_BENCH_FULL=str(tmp_path / "full.json"))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29561", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--backend", "gloo", "--share-device"], env=env, capture_output=True, text=True, timeout=900)
    objs = json_lines(r.stdout)
    assert len(objs) == 1, r.stderr[-2000:]
    j = objs[0]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline"):
        assert key in j
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak" and j["vs_baseline"] is None and "workload" in j["config"]
    assert abs(j["value"] - 2 * (1 << 20) * 3 / (j["ms_per_step"] * 3e-3)) < 1e-3 * j["value"]
    # ONE line, short enough for the driver's record, and NOTHING else on stdout: what gloo / RCCL print at C level goes to stderr (bench.py
    # points file descriptor 1 at stderr and writes its line to the original stdout at the end)
    assert r.stdout.count("\n") == 1 and r.stdout.startswith("{") and len(r.stdout) < 6000, r.stdout[:300]
    assert "rccl_ranks_seen" in j and j["prove_multi"]["verifier_accepts"] is True and j["prove_multi"]["n_gpus"] == 2
    assert j["prove_multi"]["all_ranks_same_proof"] is True and j["prove_multi"]["prove_seconds_gpu"] > 0
    t22 = j["prove_multi"]["transformer_k22"]               # configs[4]'s surrogate, here at k = 11 (two ranks share the device)
    assert t22["all_ranks_same_proof"] is True and t22["verifier_accepts"] is True and t22["prove_seconds_gpu"] > 0
    full = json.load(open(tmp_path / "full.json"))
    assert full["value"] == pytest.approx(j["value"], rel=1e-4)
    assert full["prove"]["sharded_sweeps"] >= 2 and "accum_einsum_matmul" in full["prove"]["circuit"]["circuit"]
    assert full["prove"]["sharding"] == "columns and arguments by owner"         # NTTs by columns, arguments by owner, one all-to-all for the sweep
    assert all(r["stats"]["exchange_bytes_received"] > 0 and r["stats"]["columns_transformed_here"] < r["stats"]["witness_columns"] for r in full["prove"]["per_rank"])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [8, 3])
def test_bench_contract_eight_and_three_ranks_on_one_device(hip, world, tmp_path):
    """first contact for the driver's 8-rank run, as far as one GPU allows: `bench.py --gpus N` under torchrun with N = 8
    and a world that is not a power of two, every rank on GPU 0 over gloo: ONE JSON line, the whole-job value of N ranks, the strong-scaling
    MSMs cut into N unequal point ranges, and the sharded end-to-end prove (k = 14 here: eight provers share one device) emitting the same
    proof on every rank"""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, EZKL_BENCH_PROVE_TIMEOUT="300", EZKL_BENCH_MULTI_MLP20="0", EZKL_BENCH_MULTI_K="14", EZKL_BENCH_MULTI_K22="0",
               EZKL_BENCH_FULL=str(tmp_path / "full.json"))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(29570 + world), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
                        "--backend", "gloo", "--share-device"], env=env, capture_output=True, text=True, timeout=900)
    objs = json_lines(r.stdout)
    assert len(objs) == 1, r.stderr[-2000:]
    j = objs[0]
    assert j["n_gpus"] == world and j["steps"] == 2 and j["scaling"] == "weak"
    assert abs(j["value"] - world * (1 << 20) * 2 / (j["ms_per_step"] * 2e-3)) < 1e-3 * j["value"]
    assert r.stdout.count("\n") == 1 and r.stdout.startswith("{") and len(r.stdout) < 6000, r.stdout[:300]
    assert j["msm_strong_scaling"]["2^20"]["pts_per_s"] > 0 and j["prove_multi"]["all_ranks_same_proof"] is True and "rccl_ranks_seen" in j
    full = json.load(open(tmp_path / "full.json"))
    strong = full["extra"]["msm_strong_scaling"]
    assert "error" not in strong and strong["2^20"]["points_per_rank"] in ((1 << 20) // world, (1 << 20) // world + 1)
    p = full["prove"]
    assert "error" not in p, p
    assert p["n_gpus"] == world and p["all_ranks_same_proof"] is True and p["verifier_accepts"] is True
    assert len(p["per_rank"]) == world


@pytest.mark.gpu
def test_bench_rank_failure_ends_the_job(hip):
    """a rank that dies before the first collective must end the job, not hang it: torchrun tears the other ranks down and exits non-zero
    well inside the timeout, and no JSON line is printed (EZKL_BENCH_FAIL_RANK is the test hook in bench.py)"""
    import os, subprocess, sys, time
    from conftest import ROOT
    env = dict(os.environ, EZKL_BENCH_FAIL_RANK="1", EZKL_BENCH_MULTI_K22="0")
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29569", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--backend", "gloo", "--share-device", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not json_lines(r.stdout)
    assert time.time() - t0 < 300
