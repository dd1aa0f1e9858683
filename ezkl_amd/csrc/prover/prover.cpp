// Native host prover: halo2-shaped keygen + create_proof (KZG / SHPLONK, EvmTranscript) over the C ABI of
// include/ezkl_hip.h.  Every O(n) step -- commitments (MSM), iNTT / coset NTT, the quotient sweep, grand products /
// sums, polynomial evaluation, the SHPLONK quotients -- is a kernel of libezkl_hip.so on resident columns; this file
// owns the transcript, the randomness and the O(1) scalar glue.  Round order follows SURVEY.md §3.1 ([UPSTREAM]
// halo2_proofs::plonk::prover::create_proof, called at /root/reference/src/pfsys/mod.rs:456-463): advice commits ->
// (theta, m) -> beta, gamma -> permutation products -> lookup sums -> random poly -> y -> quotient pieces -> x ->
// evaluations -> SHPLONK.  The Python restatement ezkl_amd/plonk.py is the executable spec: both emit identical bytes.
#include <algorithm>
#include <chrono>
#include <functional>
#include <map>
#include <memory>
#include <random>
#include <set>
#include <unordered_map>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "ezkl_hip.hpp"
#include "ezkl_prover.h"
#include "hostfield.hpp"
#include "transcript.hpp"
#include "pairing.hpp"

namespace ezkl_prover {
using ezkl_hip::check;
using ezkl_hip::DeviceColumn;
using ezkl_hip::Error;
using Col = std::shared_ptr<DeviceColumn>;

static thread_local std::string g_last_error;
// blinding factors: halo2's ConstraintSystem::blinding_factors() = max(3, most queries of one advice column) + 2, carried by the
// blob or derived from the queries; ezkl's circuits give 5 (/root/reference/src/graph/mod.rs:100).  The last blinding+1 rows are unusable.

// ------------------------------------------------------------------ constraint system
enum NodeOp : uint32_t { N_CONST = 0, N_ADV, N_FIX, N_INST, N_CHAL, N_NEG, N_ADD, N_SUB, N_MUL };
struct Node {
    uint32_t op, a, b;
    Fe c;
};
struct Query {
    uint32_t col;
    int32_t rot;
    bool operator<(const Query& o) const { return col != o.col ? col < o.col : rot < o.rot; }
    bool operator==(const Query& o) const { return col == o.col && rot == o.rot; }
};
struct Lookup {
    std::vector<std::vector<uint32_t>> inputs;
    std::vector<uint32_t> table;
};
// MSMs sharded by points (ezkl_prover_cs_set_shard): this rank's SRS handles hold points [lo, hi) only
struct Shard {
    uint32_t lo = 0, hi = 0;          // hi == 0: not sharded
    ezkl_fold_fn fold = nullptr;
    void* user = nullptr;
    ezkl_gather_fn gather = nullptr;  // optional: the quotient sweep sharded by rows (ezkl_prover_cs_set_sweep_gather)
    void* gather_user = nullptr;
    // optional: columns and arguments have owners (ezkl_prover_cs_set_shard_exchange)
    ezkl_allgather_host_fn allgather_host = nullptr;
    ezkl_exchange_fn exchange = nullptr;
    void* xuser = nullptr;
    mutable uint64_t sharded_sweeps = 0;
    mutable uint64_t stats[4] = {0, 0, 0, 0};     // ezkl_prover_cs_shard_stats
    // the SRS handles hold ALL 2^k points on every rank (ezkl_prover_cs_set_shard_full_bases; 288 GB of HBM per GPU: a 2^22 base set
    // with its window tables is 3.5 GB): a commit batch is then divided by COLUMNS -- whole MSMs, which keep the per-call tail of a
    // 2^k-point MSM off the critical path instead of paying it on every 2^k / world slice -- and by point ranges inside a column
    // only when the batch has fewer columns than there are ranks
    bool full_bases = false;
    bool on() const { return hi != 0; }
    // equal power-of-two slices: rank / log2(world) of this one, or false
    bool geometry(uint32_t n, uint32_t& rank, uint32_t& log_world) const {
        const uint32_t len = hi - lo;
        if (!on() || len == 0 || n % len || lo % len) return false;
        const uint32_t world = n / len;
        if (world & (world - 1)) return false;
        rank = lo / len;
        log_world = 0;
        while ((1u << log_world) < world) log_world++;
        return true;
    }
};
// Who does what in one proof.  One rank (or a sharded prover without the exchange callbacks): everything is mine.  Owner mode: witness
// column / argument number i belongs to rank i mod world; only its owner computes, transforms and commits it.
struct Topo {
    uint32_t world = 1, rank = 0, log_world = 0;
    bool owners = false;
    bool mine(size_t i) const { return !owners || (uint32_t)(i % world) == rank; }
    uint32_t owner(size_t i) const { return owners ? (uint32_t)(i % world) : rank; }
};
struct ConstraintSystem {
    uint32_t k = 0, n = 0, n_advice = 0, n_fixed = 0, n_instance = 0, n_challenges = 0;
    std::vector<uint32_t> advice_phase;
    std::vector<Node> nodes;
    std::vector<uint32_t> gates;
    std::vector<std::pair<uint32_t, uint32_t>> perm;      // (kind = N_ADV | N_FIX | N_INST, col)
    std::vector<Lookup> lookups;
    uint32_t usable = 0, degree = 0, chunk = 0, ext_k = 0, n_chunks = 0;
    uint32_t blinding = 0, minimum_degree = 0;           // 0 = derive / none (blob version 1)
    uint32_t n_selectors = 0;                             // halo2 selectors behind the fixed columns: sizes the selector section of vk / pk files
    bool queries_given = false;                           // halo2's order of first query (blob version 2)
    bool advice_by_pointer = false;                       // ezkl_prover_cs_set_advice_by_pointer
    std::vector<uint8_t> unblinded;                       // per advice column: unusable rows hold Blind::default() = 1
    std::array<uint8_t, 32> blob_hash{};                  // keccak256 of the blob: binds gates / lookups / queries into the vk digest
    std::vector<Query> advice_queries, fixed_queries, instance_queries;
    std::vector<uint32_t> deg_memo;
    Shard shard;

    uint32_t deg(uint32_t id) {
        if (deg_memo[id] != UINT32_MAX) return deg_memo[id];
        const Node& nd = nodes[id];
        uint32_t d;
        switch (nd.op) {
        case N_CONST: case N_CHAL: d = 0; break;
        case N_ADV: case N_FIX: case N_INST: d = 1; break;
        case N_NEG: d = deg(nd.a); break;
        case N_ADD: case N_SUB: d = std::max(deg(nd.a), deg(nd.b)); break;
        default: d = deg(nd.a) + deg(nd.b); break;
        }
        return deg_memo[id] = d;
    }
    void collect(uint32_t id, std::set<Query> out[3], std::vector<uint8_t>& seen) const {
        if (seen[id]) return;
        seen[id] = 1;
        const Node& nd = nodes[id];
        if (nd.op == N_ADV || nd.op == N_FIX || nd.op == N_INST) out[nd.op - N_ADV].insert(Query{nd.a, (int32_t)nd.b});
        else if (nd.op == N_NEG) collect(nd.a, out, seen);
        else if (nd.op >= N_ADD) { collect(nd.a, out, seen); collect(nd.b, out, seen); }
    }
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> perm_chunks() const {
        std::vector<std::vector<std::pair<uint32_t, uint32_t>>> out;
        for (size_t i = 0; i < perm.size(); i += chunk) out.emplace_back(perm.begin() + i, perm.begin() + std::min(perm.size(), i + chunk));
        return out;
    }
    void finalize() {
        n = 1u << k;
        deg_memo.assign(nodes.size(), UINT32_MAX);
        uint32_t d = 3;
        for (uint32_t g : gates) d = std::max(d, deg(g));
        for (auto& l : lookups) {                          // l_active * phi * prod(f_j + beta) * (t + beta)
            uint32_t s = 2, tmax = 0;
            for (auto& t : l.inputs) {
                uint32_t m = 0;
                for (uint32_t e : t) m = std::max(m, deg(e));
                s += m;
            }
            for (uint32_t e : l.table) tmax = std::max(tmax, deg(e));
            d = std::max(d, s + tmax);
        }
        degree = d = std::max(d, minimum_degree);
        chunk = d - 2;
        ext_k = k;
        while ((1ull << ext_k) < (uint64_t)n * (d - 1)) ext_k++;
        std::set<Query> qs[3];
        std::vector<uint8_t> seen(nodes.size(), 0);
        for (uint32_t g : gates) collect(g, qs, seen);
        for (auto& pc : perm) qs[pc.first - N_ADV].insert(Query{pc.second, 0});
        for (auto& l : lookups) {
            for (auto& t : l.inputs)
                for (uint32_t e : t) collect(e, qs, seen);
            for (uint32_t e : l.table) collect(e, qs, seen);
        }
        if (queries_given) {                              // must cover what the expressions read, without duplicates
            std::vector<Query>* given[3] = {&advice_queries, &fixed_queries, &instance_queries};
            for (int t = 0; t < 3; t++) {
                std::set<Query> have(given[t]->begin(), given[t]->end());
                if (have.size() != given[t]->size()) throw Error(EZKL_ERR_INVALID, "duplicate query");
                for (auto& q : qs[t])
                    if (!have.count(q)) throw Error(EZKL_ERR_INVALID, "query lists do not cover the expressions");
            }
        } else {
            advice_queries.assign(qs[0].begin(), qs[0].end());
            fixed_queries.assign(qs[1].begin(), qs[1].end());
            instance_queries.assign(qs[2].begin(), qs[2].end());
        }
        if (blinding == 0) {                              // halo2 ConstraintSystem::blinding_factors
            std::map<uint32_t, uint32_t> per_col;
            uint32_t most = 1;
            for (auto& q : advice_queries) most = std::max(most, ++per_col[q.col]);
            blinding = std::max(3u, most) + 2;
        }
        if (blinding + 2 > n) throw Error(EZKL_ERR_INVALID, "no usable rows");
        usable = n - blinding - 1;
        n_chunks = perm.empty() ? 0 : (uint32_t)((perm.size() + chunk - 1) / chunk);
    }
};

struct Reader {
    const uint8_t* p;
    size_t left;
    uint32_t u32() {
        if (left < 4) throw Error(EZKL_ERR_INVALID, "constraint system blob truncated");
        uint32_t v;
        std::memcpy(&v, p, 4);
        p += 4; left -= 4;
        return v;
    }
    void bytes(void* out, size_t m) {
        if (left < m) throw Error(EZKL_ERR_INVALID, "constraint system blob truncated");
        std::memcpy(out, p, m);
        p += m; left -= m;
    }
};
static void invalid(bool cond, const char* what) {
    if (cond) throw Error(EZKL_ERR_INVALID, what);
}
static std::unique_ptr<ConstraintSystem> parse_cs(const void* blob, size_t len) {
    Reader r{(const uint8_t*)blob, len};
    invalid(r.u32() != 0x53435a45u, "bad magic");
    const uint32_t version = r.u32();
    invalid(version != 1 && version != 2, "unsupported version");
    auto cs = std::make_unique<ConstraintSystem>();
    cs->blob_hash = keccak256((const uint8_t*)blob, len);
    cs->k = r.u32(); cs->n_advice = r.u32(); cs->n_fixed = r.u32(); cs->n_instance = r.u32(); cs->n_challenges = r.u32();
    invalid(cs->k < 4 || cs->k > 28, "k out of range");
    invalid(cs->n_advice > (1u << 16) || cs->n_fixed > (1u << 16) || cs->n_instance > (1u << 16) || cs->n_challenges > (1u << 16), "column count out of range");
    for (uint32_t i = 0; i < cs->n_advice; i++) {
        cs->advice_phase.push_back(r.u32());
        invalid(cs->advice_phase.back() > 1, "advice phase must be 0 or 1");
    }
    cs->unblinded.assign(cs->n_advice, 0);
    if (version >= 2) {
        cs->blinding = r.u32();
        cs->minimum_degree = r.u32();
        invalid(cs->blinding > 64 || cs->minimum_degree > 64, "blinding / minimum degree out of range");
        const uint32_t nu = r.u32();
        invalid((size_t)nu * 4 > r.left, "unblinded list truncated");
        for (uint32_t i = 0; i < nu; i++) {
            const uint32_t c = r.u32();
            invalid(c >= cs->n_advice, "unblinded column out of range");
            cs->unblinded[c] = 1;
        }
        cs->n_selectors = r.u32();
        invalid(cs->n_selectors > (1u << 20), "selector count out of range");
    }
    const uint32_t nn = r.u32();
    invalid((size_t)nn * 48 > r.left, "node table truncated");
    for (uint32_t i = 0; i < nn; i++) {
        Node nd;
        nd.op = r.u32(); nd.a = r.u32(); nd.b = r.u32();
        r.u32();
        r.bytes(nd.c.v.data(), 32);
        invalid(nd.op > N_MUL, "bad node op");
        if (nd.op == N_CONST) invalid(cmp(nd.c.v, FR.p) >= 0, "non-canonical constant");
        if (nd.op == N_ADV) invalid(nd.a >= cs->n_advice, "advice column out of range");
        if (nd.op == N_FIX) invalid(nd.a >= cs->n_fixed, "fixed column out of range");
        if (nd.op == N_INST) invalid(nd.a >= cs->n_instance, "instance column out of range");
        if (nd.op == N_CHAL) invalid(nd.a >= cs->n_challenges, "challenge index out of range");
        if (nd.op >= N_NEG) invalid(nd.a >= i, "child must precede parent");
        if (nd.op >= N_ADD) invalid(nd.b >= i, "child must precede parent");
        cs->nodes.push_back(nd);
    }
    auto node_list = [&](std::vector<uint32_t>& out) {
        const uint32_t m = r.u32();
        invalid((size_t)m * 4 > r.left, "list truncated");
        for (uint32_t i = 0; i < m; i++) {
            out.push_back(r.u32());
            invalid(out.back() >= nn, "node id out of range");
        }
    };
    node_list(cs->gates);
    const uint32_t np = r.u32();
    invalid((size_t)np * 8 > r.left, "permutation list truncated");
    for (uint32_t i = 0; i < np; i++) {
        uint32_t kind = r.u32(), col = r.u32();
        invalid(kind < N_ADV || kind > N_INST, "bad permutation column kind");
        invalid(col >= (kind == N_ADV ? cs->n_advice : kind == N_FIX ? cs->n_fixed : cs->n_instance), "permutation column out of range");
        cs->perm.emplace_back(kind, col);
    }
    const uint32_t nl = r.u32();
    for (uint32_t i = 0; i < nl; i++) {
        Lookup l;
        const uint32_t ni = r.u32();
        invalid(ni == 0 || (size_t)ni * 4 > r.left, "lookup without inputs");
        for (uint32_t j = 0; j < ni; j++) {
            l.inputs.emplace_back();
            node_list(l.inputs.back());
            invalid(l.inputs.back().empty(), "empty lookup tuple");
        }
        node_list(l.table);
        for (auto& t : l.inputs) invalid(t.size() != l.table.size(), "lookup arity mismatch");
        cs->lookups.push_back(std::move(l));
    }
    if (version >= 2) {
        cs->queries_given = r.u32() != 0;
        if (cs->queries_given) {
            std::vector<Query>* lists[3] = {&cs->advice_queries, &cs->fixed_queries, &cs->instance_queries};
            const uint32_t limits[3] = {cs->n_advice, cs->n_fixed, cs->n_instance};
            for (int t = 0; t < 3; t++) {
                const uint32_t m = r.u32();
                invalid((size_t)m * 8 > r.left, "query list truncated");
                for (uint32_t i = 0; i < m; i++) {
                    Query q;
                    q.col = r.u32();
                    q.rot = (int32_t)r.u32();
                    invalid(q.col >= limits[t], "query column out of range");
                    lists[t]->push_back(q);
                }
            }
        }
    }
    invalid(r.left != 0, "trailing bytes");
    cs->finalize();
    return cs;
}

// ------------------------------------------------------------------ gate programs (GraphEvaluator)
struct Src {
    uint32_t kind, idx, rot;
};
struct Program {
    uint32_t k, ext_k;
    std::vector<uint32_t> code;
    std::vector<U256> constants;
    std::vector<int32_t> rotations;
    uint32_t n_int = 0;
    Program(uint32_t k_, uint32_t e_) : k(k_), ext_k(e_) {}
    Src constant(const Fe& c) {
        for (size_t i = 0; i < constants.size(); i++)
            if (constants[i] == c.v) return Src{EZKL_SRC_CONST, (uint32_t)i, 0};
        constants.push_back(c.v);
        return Src{EZKL_SRC_CONST, (uint32_t)constants.size() - 1, 0};
    }
    uint32_t rotation(int32_t r) {
        for (size_t i = 0; i < rotations.size(); i++)
            if (rotations[i] == r) return (uint32_t)i;
        rotations.push_back(r);
        return (uint32_t)rotations.size() - 1;
    }
    Src column(uint32_t idx, int32_t rot = 0) { return Src{EZKL_SRC_COLUMN, idx, rotation(rot)}; }
    Src challenge(uint32_t idx) const { return Src{EZKL_SRC_CHALLENGE, idx, 0}; }
    Src previous() const { return Src{EZKL_SRC_PREVIOUS, 0, 0}; }
    Src calc(uint32_t op, Src s0, Src s1 = Src{EZKL_SRC_CONST, 0, 0}, int64_t target = -1) {
        const uint32_t t = target < 0 ? n_int++ : (uint32_t)target;
        const uint32_t w[8] = {op, t, s0.kind, s0.idx, s0.rot, s1.kind, s1.idx, s1.rot};
        code.insert(code.end(), w, w + 8);
        return Src{EZKL_SRC_INTERMEDIATE, t, 0};
    }
    Src add(Src a, Src b) { return calc(EZKL_OP_ADD, a, b); }
    Src sub(Src a, Src b) { return calc(EZKL_OP_SUB, a, b); }
    Src mul(Src a, Src b) { return calc(EZKL_OP_MUL, a, b); }
    Src horner(Src start, const std::vector<Src>& parts, Src factor) {
        Src t = calc(EZKL_OP_STORE, start);
        for (auto& p : parts) calc(EZKL_OP_HORNER_STEP, p, factor, t.idx);
        return t;
    }
    // The program of ONE row shard of a sweep split over 2^log_world ranks: every distinct (column, rotation) the code reads
    // becomes a column of its own at rotation 0; windows[j] = (column, row shift) says which rows of the original extended
    // column the j-th new column holds: [lo + shift, hi + shift) mod 2^ext_k for the shard [lo, hi).  The shard is a domain of
    // 2^(ext_k - log_world) rows, so ezkl_hip_eval_h_dev runs it unchanged (GraphProgram.row_sharded is the Python twin).
    Program row_sharded(uint32_t log_world, std::vector<std::pair<uint32_t, int64_t>>& windows) const {
        Program sub(k - log_world, ext_k - log_world);
        sub.constants = constants;
        sub.n_int = n_int;
        sub.rotations = {0};
        sub.code = code;
        const int64_t step = (int64_t)1 << (ext_k - k);
        std::map<std::pair<uint32_t, int64_t>, uint32_t> index;
        for (size_t i = 0; i < sub.code.size(); i += 8)
            for (size_t s : {(size_t)2, (size_t)5}) {
                if (sub.code[i + s] != EZKL_SRC_COLUMN) continue;
                const std::pair<uint32_t, int64_t> key{sub.code[i + s + 1], rotations[sub.code[i + s + 2]] * step};
                auto it = index.find(key);
                if (it == index.end()) {
                    it = index.emplace(key, (uint32_t)windows.size()).first;
                    windows.push_back(key);
                }
                sub.code[i + s + 1] = it->second;
                sub.code[i + s + 2] = 0;
            }
        return sub;
    }
    void run(const std::vector<Col>& cols, const std::vector<Fe>& chal, void* out) const {
        std::vector<const void*> ptrs;
        for (auto& c : cols) ptrs.push_back(c->ptr());
        run_ptrs(ptrs, chal, out);
    }
    // compile / load the kernel without running it (ezkl_hip_eval_h_prepare): nothing is read through the column pointers
    void prepare(size_t n_cols, size_t n_chal) const {
        std::vector<const void*> ptrs(n_cols ? n_cols : 1, nullptr);
        std::vector<Fe> chal(n_chal ? n_chal : 1, Fe::zero());
        ezkl_program_t p{};
        p.code = code.data();
        p.n_instr = (uint32_t)(code.size() / 8);
        p.n_intermediates = n_int;
        p.constants = constants.data();
        p.n_constants = (uint32_t)constants.size();
        p.rotations = rotations.data();
        p.n_rotations = (uint32_t)rotations.size();
        p.columns = ptrs.data();
        p.n_columns = (uint32_t)n_cols;
        p.challenges = chal.data();
        p.n_challenges = (uint32_t)n_chal;
        p.k = k;
        p.ext_k = ext_k;
        check(ezkl_hip_eval_h_prepare(&p), "ezkl_hip_eval_h_prepare");
    }
    void run_ptrs(std::vector<const void*> ptrs, const std::vector<Fe>& chal, void* out) const {
        const uint32_t n_cols = (uint32_t)ptrs.size();
        if (ptrs.empty()) ptrs.push_back(nullptr);
        ezkl_program_t p{};
        p.code = code.data();
        p.n_instr = (uint32_t)(code.size() / 8);
        p.n_intermediates = n_int;
        p.constants = constants.data();
        p.n_constants = (uint32_t)constants.size();
        p.rotations = rotations.data();
        p.n_rotations = (uint32_t)rotations.size();
        p.columns = ptrs.data();
        p.n_columns = n_cols;
        p.challenges = chal.data();
        p.n_challenges = (uint32_t)chal.size();
        p.k = k;
        p.ext_k = ext_k;
        check(ezkl_hip_eval_h_dev(&p, out, nullptr), "ezkl_hip_eval_h_dev");
    }
};
// emit an expression into a program; col_slot(kind, col) -> column slot, chal_src(idx) -> source of user challenge idx
struct Lowering {
    const ConstraintSystem& cs;
    Program& prog;
    std::function<uint32_t(uint32_t, uint32_t)> col_slot;
    std::function<Src(uint32_t)> chal_src;
    std::unordered_map<uint32_t, Src> memo;
    Src lower(uint32_t id) {
        auto it = memo.find(id);
        if (it != memo.end()) return it->second;
        const Node& nd = cs.nodes[id];
        Src r;
        switch (nd.op) {
        case N_CONST: r = prog.constant(nd.c); break;
        case N_CHAL: r = chal_src(nd.a); break;
        case N_ADV: case N_FIX: case N_INST: r = prog.column(col_slot(nd.op, nd.a), (int32_t)nd.b); break;
        case N_NEG: r = prog.calc(EZKL_OP_NEGATE, lower(nd.a)); break;
        default: {
            Src a = lower(nd.a), b = lower(nd.b);
            r = prog.calc(nd.op == N_ADD ? EZKL_OP_ADD : nd.op == N_SUB ? EZKL_OP_SUB : EZKL_OP_MUL, a, b);
        } break;
        }
        memo[id] = r;
        return r;
    }
    // theta-compression of a tuple of expressions: ((e0*theta + e1)*theta + e2)...  (halo2 compress_expressions)
    Src compress(const std::vector<uint32_t>& tuple, Src theta) {
        Src acc = lower(tuple[0]);
        for (size_t i = 1; i < tuple.size(); i++) {
            Src m = prog.mul(acc, theta);
            acc = prog.add(m, lower(tuple[i]));
        }
        return acc;
    }
};

// ------------------------------------------------------------------ resident-column helpers (each = one or a few C-ABI calls)
struct Backend {
    uint32_t k, n;
    ezkl_bases_t g, gl;
    Shard shard;
    Topo topo;
    Fe one = Fe::one();
    uint64_t* recv_bytes = nullptr;       // the constraint system's counter of bytes received through the exchange (create_proof sets it)
    Backend(uint32_t k_, uint32_t n_, ezkl_bases_t g_, ezkl_bases_t gl_, const Shard& sh = Shard()) : k(k_), n(n_), g(g_), gl(gl_), shard(sh) {
        uint32_t r = 0, lw = 0;
        if (shard.on() && shard.geometry(n, r, lw)) {
            topo.world = 1u << lw; topo.rank = r; topo.log_world = lw;
            // owner mode needs complete base sets on every rank (whole MSMs by the owner) and the exchange callbacks
            topo.owners = topo.world > 1 && shard.full_bases && shard.exchange && shard.allgather_host && shard.gather;
        }
    }
    Backend(const Backend&) = delete;
    ~Backend() {
        if (aux) {                             // also on unwinding: nothing queued on the aux stream may outlive its columns
            (void)ezkl_hip_stream_synchronize(aux);   // the stream itself belongs to the context
        }
    }
    std::vector<Col> aux_keep;                 // inputs / outputs of work queued on the aux stream, alive until the stream is drained
    // A second stream for the NTTs of finished columns: a column's coefficient and extended-coset forms do not depend on any
    // challenge, so they are queued the moment the column is final and run in the shadow of the commit phases (the advice phase
    // is PCIe-bound, single MSMs leave the GPU half empty during their sort / reduce tails); step 7 only waits for the stream.
    void* aux = nullptr;
    void* aux_stream() {
        if (!aux) check(ezkl_hip_context_stream(&aux), "ezkl_hip_context_stream");
        return aux;
    }
    void aux_sync() {
        if (aux) check(ezkl_hip_stream_synchronize(aux), "ezkl_hip_stream_synchronize");
        aux_keep.clear();
    }
    struct Forms {
        Col poly, coset;
    };
    // lagrange -> (coefficients, extended cosets in COSET-MAJOR order: element b n + j = p(zeta w_ext^b omega^j)), stream-ordered on the aux
    // stream; `lagrange` must stay untouched until aux_sync()
    // stream_cosets (the degraded mode of a key that does not fit with its extended columns, ProvingKey::stream): no column keeps its
    // extended form; the sweep rebuilds coset b of every column from the coefficients when it gets to unit b (coset_of)
    bool stream_cosets = false;
    Forms forms_alloc(uint32_t ext_k) const { return Forms{alloc(n), stream_cosets ? Col() : alloc((size_t)1 << ext_k)}; }
    // coset b of coeff_to_extended(h): out[j] = h(zeta w_ext^b omega^j), n rows
    Col coset_of(const Col& h, uint32_t ext_k, uint32_t b) const {
        Col o = alloc(n);
        check(ezkl_hip_coeff_to_cosets_range_dev(h->ptr(), o->ptr(), 1, n, n, k, ext_k, b, 1, nullptr), "ezkl_hip_coeff_to_cosets_range_dev");
        return o;
    }
    Forms forms_async(const Col& lagrange, uint32_t ext_k, const Forms* pre = nullptr) {
        Forms f = pre ? *pre : forms_alloc(ext_k);
        aux_keep.insert(aux_keep.end(), {lagrange, f.poly, f.coset});
        void* st = aux_stream();
        // the column was finished by calls on the library stream, which create_proof runs asynchronously (ezkl_hip_set_async): the
        // auxiliary stream starts behind everything queued there so far -- which also covers any earlier reader of the recycled
        // blocks the forms were just allocated from
        check(ezkl_hip_stream_wait_library(st), "ezkl_hip_stream_wait_library");
        const Fe winv = omega(k).inv();
        check(ezkl_hip_vec_scale_dev(lagrange->ptr(), one.v.data(), f.poly->ptr(), n, st), "ezkl_hip_vec_scale_dev");
        check(ezkl_hip_ntt_dev(f.poly->ptr(), k, winv.v.data(), 1, 1, n, st), "ezkl_hip_ntt_dev");
        if (f.coset) check(ezkl_hip_coeff_to_cosets_dev(f.poly->ptr(), f.coset->ptr(), 1, n, (size_t)1 << ext_k, k, ext_k, st), "ezkl_hip_coeff_to_cosets_dev");
        return f;
    }
    size_t commit_first() const { return shard.on() ? shard.lo : 0; }
    size_t commit_count() const { return shard.on() ? shard.hi - shard.lo : n; }
    // sharded: the partial sums over this rank's slice become the sums over all ranks (one all_gather per batch on the caller's side)
    void fold(std::vector<G1>& pts) const {
        if (!shard.on() || pts.empty()) return;
        if (shard.fold(shard.user, pts.data(), (uint32_t)pts.size()) != 0) throw Error(EZKL_ERR_INVALID, "fold callback failed");
    }

    Col alloc(size_t m) const { return std::make_shared<DeviceColumn>(m); }
    Col upload(const void* host, size_t m) const {
        Col c = alloc(m);
        check(ezkl_hip_memcpy_h2d(c->ptr(), host, m * 32), "ezkl_hip_memcpy_h2d");
        return c;
    }
    Col upload(const std::vector<U256>& v) const { return upload(v.data(), v.size()); }
    std::vector<U256> download(const Col& c, size_t m) const {
        std::vector<U256> v(m);
        check(ezkl_hip_memcpy_d2h(v.data(), c->ptr(), m * 32), "ezkl_hip_memcpy_d2h");
        return v;
    }
    static void* at(const Col& c, size_t off) { return (uint8_t*)c->ptr() + 32 * off; }
    void scale_into(const void* src, const Fe& s, void* dst, size_t m) const { check(ezkl_hip_vec_scale_dev(src, s.v.data(), dst, m, nullptr), "ezkl_hip_vec_scale_dev"); }
    void vec(int op, const void* a, const void* b, void* o, size_t m) const { check(ezkl_hip_vec_op_dev(op, a, b, o, m, nullptr), "ezkl_hip_vec_op_dev"); }
    void fill(void* dst, const Fe& v, size_t m) const { check(ezkl_hip_vec_fill_dev(dst, v.v.data(), m, nullptr), "ezkl_hip_vec_fill_dev"); }
    void scan(int op, bool exclusive, const void* in, void* out, size_t m) const { check(ezkl_hip_prefix_scan_dev(op, exclusive ? 1 : 0, in, out, m, nullptr), "ezkl_hip_prefix_scan_dev"); }
    void invert(void* a, size_t m) const { check(ezkl_hip_batch_invert_dev(a, m, nullptr), "ezkl_hip_batch_invert_dev"); }
    Col clone(const Col& h) const {
        Col o = alloc(h->len());
        scale_into(h->ptr(), one, o->ptr(), h->len());
        return o;
    }
    Col zeros(size_t m) const {
        Col o = alloc(m);
        fill(o->ptr(), Fe::zero(), m);
        return o;
    }
    // the n-row indicator of rows [lo, hi), filled on the device (l0, l_last, l_active_row, the coefficients of X: a 32 MiB host
    // vector per column and its pageable upload cost ~20 ms each at k = 20 -- a fifth of the key load of a one-shot prove)
    Col indicator(size_t lo, size_t hi) const {
        Col o = zeros(n);
        if (hi > lo) fill(at(o, lo), one, hi - lo);
        return o;
    }
    // small = witness-shaped columns (advice, multiplicities): the batch runs as fused groups (ezkl_hip_msm_g1_batch_small_dev)
    int msm_batch(ezkl_bases_t b, size_t first, const void* const* ptrs, size_t m, size_t len, void* out, bool small) const {
        return small ? ezkl_hip_msm_g1_batch_small_dev(b, first, ptrs, m, len, out, nullptr) : ezkl_hip_msm_g1_batch_dev(b, first, ptrs, m, len, out, nullptr);
    }
    std::vector<G1> commit_with(ezkl_bases_t b, const std::vector<Col>& hs, bool small = false) const {
        std::vector<G1> out(hs.size());                      // zero bytes = the identity: what a rank contributes for work it does not do
        if (hs.empty()) return out;
        uint32_t rank = 0, log_world = 0;
        if (shard.on() && shard.full_bases && shard.geometry(n, rank, log_world)) {
            const uint32_t world = 1u << log_world, m = (uint32_t)hs.size();
            if (m >= world) {                                // by columns: column i on rank i mod world, whole MSMs
                std::vector<const void*> ptrs;
                std::vector<uint32_t> mine;
                for (uint32_t i = rank; i < m; i += world) { ptrs.push_back(hs[i]->ptr()); mine.push_back(i); }
                std::vector<G1> part(ptrs.size());
                check(msm_batch(b, 0, ptrs.data(), ptrs.size(), n, part.data(), small), "ezkl_hip_msm_g1_batch_dev");
                for (size_t j = 0; j < mine.size(); j++) out[mine[j]] = part[j];
            } else {                                         // fewer columns than ranks: 2^t ranks per column, each a point range
                uint32_t pieces = 1;
                while (pieces * 2 * m <= world) pieces *= 2;
                if (rank < m * pieces) {
                    const uint32_t col = rank / pieces, q = rank % pieces;
                    const size_t len = n / pieces, first = (size_t)q * len;
                    const void* ptr = at(hs[col], first);
                    check(ezkl_hip_msm_g1_batch_dev(b, first, &ptr, 1, len, &out[col], nullptr), "ezkl_hip_msm_g1_batch_dev");
                }
            }
            fold(out);
            return out;
        }
        std::vector<const void*> ptrs;
        for (auto& h : hs) ptrs.push_back(at(h, commit_first()));
        check(msm_batch(b, shard.on() && shard.full_bases ? commit_first() : 0, ptrs.data(), ptrs.size(), commit_count(), out.data(), small),
              "ezkl_hip_msm_g1_batch_dev");
        fold(out);
        return out;
    }
    std::vector<G1> commit_lagrange(const std::vector<Col>& hs, bool small = false) const { return commit_with(gl, hs, small); }
    std::vector<G1> commit(const std::vector<Col>& hs) const { return commit_with(g, hs); }
    // owner mode: hs[i] is a column on its owner and null elsewhere; every rank commits what it holds (whole MSMs), the identity for
    // the rest, and the fold sums the partials -- the commitments of the batch on every rank
    std::vector<G1> commit_owned(ezkl_bases_t b, const std::vector<Col>& hs, bool small) const {
        std::vector<G1> out(hs.size());
        std::vector<const void*> ptrs;
        std::vector<size_t> where;
        for (size_t i = 0; i < hs.size(); i++)
            if (hs[i]) { ptrs.push_back(hs[i]->ptr()); where.push_back(i); }
        if (!ptrs.empty()) {
            std::vector<G1> part(ptrs.size());
            check(msm_batch(b, 0, ptrs.data(), ptrs.size(), n, part.data(), small), "ezkl_hip_msm_g1_batch_dev");
            for (size_t j = 0; j < where.size(); j++) out[where[j]] = part[j];
        }
        fold(out);
        return out;
    }
    // a batch of witness columns in whichever mode the prover runs: by owner, or replicated (commit_with: one rank, by points, by columns)
    std::vector<G1> commit_columns(ezkl_bases_t b, const std::vector<Col>& hs, bool small) const {
        return topo.owners ? commit_owned(b, hs, small) : commit_with(b, hs, small);
    }
    // owner mode: the commitment of the SUM over ranks of a per-rank partial polynomial (SHPLONK's h and L: linear in the polynomials
    // each rank owns).  A reduce-scatter by point ranges: rank r receives rows [r n / world, (r + 1) n / world) of every other rank's
    // partial through the exchange (one segment per peer and direction: 32 n / world bytes), adds them to its own, commits that slice
    // against the same slice of the base set, and the fold adds the world points -- an n / world-point MSM per rank instead of a whole one
    // on every rank.  EZKL_PROVER_NO_SUM_SCATTER=1: every rank commits its whole partial (the round-3 form), same point.
    G1 commit_sum(ezkl_bases_t b, const Col& h) const {
        if (!topo.owners) return commit_with(b, {h})[0];
        static const bool scatter = getenv("EZKL_PROVER_NO_SUM_SCATTER") == nullptr;
        std::vector<G1> out(1);
        const uint32_t W = topo.world;
        if (scatter && W > 1 && n % W == 0) {
            const size_t len = n / W;
            Col pieces = alloc(len * (W - 1));                  // the peers' rows of my range, in rank order
            std::vector<ezkl_comm_seg_t> sends, recvs;
            std::vector<const void*> ptrs = {at(h, (size_t)topo.rank * len)};
            for (uint32_t p_ = 0; p_ < W; p_++) {
                if (p_ == topo.rank) continue;
                void* dst = at(pieces, (size_t)(p_ < topo.rank ? p_ : p_ - 1) * len);
                sends.push_back({(int)p_, at(h, (size_t)p_ * len), len * 32});
                recvs.push_back({(int)p_, dst, len * 32});
                ptrs.push_back(dst);
            }
            check(ezkl_hip_synchronize(), "ezkl_hip_synchronize");                 // the partial is complete on every stream
            if (shard.exchange(shard.xuser, sends.data(), sends.size(), recvs.data(), recvs.size()) != 0) throw Error(EZKL_ERR_INVALID, "exchange callback failed");
            if (recv_bytes) *recv_bytes += (uint64_t)(W - 1) * len * 32;
            const std::vector<U256> ones(W, Fe::one().v);
            Col sum = alloc(len);
            check(ezkl_hip_lincomb_dev(ptrs.data(), ones.data(), W, sum->ptr(), len, 0, nullptr), "ezkl_hip_lincomb_dev");
            const void* ptr = sum->ptr();
            check(ezkl_hip_msm_g1_batch_dev(b, (size_t)topo.rank * len, &ptr, 1, len, out.data(), nullptr), "ezkl_hip_msm_g1_batch_dev");
            fold(out);
            return out[0];
        }
        const void* ptr = h->ptr();
        check(msm_batch(b, 0, &ptr, 1, n, out.data(), false), "ezkl_hip_msm_g1_batch_dev");
        fold(out);
        return out[0];
    }
    // all_gather of `per` field elements per rank (host): v[r * per + i] valid on rank r going in, everywhere coming out
    void allgather_fe(std::vector<Fe>& v, size_t per) const {
        if (!topo.owners || per == 0) return;
        if (shard.allgather_host(shard.xuser, v.data(), per * sizeof(Fe)) != 0) throw Error(EZKL_ERR_INVALID, "allgather callback failed");
    }
    Col lagrange_to_coeff(const Col& h) const {
        Col o = clone(h);
        const Fe winv = omega(k).inv();
        check(ezkl_hip_ntt_dev(o->ptr(), k, winv.v.data(), 1, 1, n, nullptr), "ezkl_hip_ntt_dev");
        return o;
    }
    // The extended domain lives in COSET-MAJOR order everywhere in this prover (ezkl_hip_coeff_to_cosets_dev): E = 2^(ext_k - k) cosets
    // of n rows, coset b = {zeta w_ext^b omega^j}.  A rotation is a shift inside a coset, the vanishing polynomial is a constant on a
    // coset, and a coset (or a row range of one) is the unit of the quotient sweep.
    Col coeff_to_extended(const Col& h, uint32_t ext_k) const {
        Col o = alloc((size_t)1 << ext_k);
        check(ezkl_hip_coeff_to_cosets_dev(h->ptr(), o->ptr(), 1, n, (size_t)1 << ext_k, k, ext_k, nullptr), "ezkl_hip_coeff_to_cosets_dev");
        return o;
    }
    // the cosets of the extended domain whose rows THIS rank sweeps (create_proof's units): all of them unless columns have owners; by
    // owner, world <= E: E / world whole cosets; world > E: the one coset its row range lies in
    void key_range(uint32_t ext_k, uint32_t& first, uint32_t& count) const {
        const uint32_t E = 1u << (ext_k - k);
        first = 0; count = E;
        if (!topo.owners) return;
        if (topo.world <= E) { count = E / topo.world; first = topo.rank * count; }
        else { count = 1; first = topo.rank / (topo.world / E); }
    }
    // coeff_to_extended restricted to key_range: out[(b - first) n + j]
    Col key_cosets(const Col& h, uint32_t ext_k) const {
        uint32_t first, count;
        key_range(ext_k, first, count);
        if (count == (1u << (ext_k - k))) return coeff_to_extended(h, ext_k);
        Col o = alloc((size_t)count * n);
        check(ezkl_hip_coeff_to_cosets_range_dev(h->ptr(), o->ptr(), 1, n, (size_t)count * n, k, ext_k, first, count, nullptr), "ezkl_hip_coeff_to_cosets_range_dev");
        return o;
    }
    // coset-major evaluations -> the 2^ext_k coefficients (natural order): transposed into the natural order of the extended domain,
    // then EvaluationDomain::extended_to_coeff as one inverse transform (one column per proof: h)
    Col extended_to_coeff(const Col& h, uint32_t ext_k) const {
        Col o = alloc((size_t)1 << ext_k);
        if (ext_k == k) scale_into(h->ptr(), one, o->ptr(), n);
        else check(ezkl_hip_cosets_transpose_dev(h->ptr(), o->ptr(), k, ext_k, 1, nullptr), "ezkl_hip_cosets_transpose_dev");
        check(ezkl_hip_coset_ntt_dev(o->ptr(), o->ptr(), 1, (size_t)1 << ext_k, (size_t)1 << ext_k, k, ext_k, 1, nullptr), "ezkl_hip_coset_ntt_dev");
        return o;
    }
    // natural <-> coset-major order of one extended column (key files hold halo2's natural order)
    Col cosets_reorder(const Col& h, uint32_t ext_k, bool to_natural) const {
        Col o = alloc((size_t)1 << ext_k);
        if (ext_k == k) scale_into(h->ptr(), one, o->ptr(), n);
        else check(ezkl_hip_cosets_transpose_dev(h->ptr(), o->ptr(), k, ext_k, to_natural ? 1 : 0, nullptr), "ezkl_hip_cosets_transpose_dev");
        return o;
    }
    // 1 / Z_H on coset b: Z_H(c_b omega^j) = c_b^n - 1, c_b = zeta w_ext^b (EvaluationDomain::divide_by_vanishing_poly's t_evaluations)
    Fe vanishing_inv(uint32_t ext_k, uint32_t b) const {
        const Fe cb = Fe{FR_ZETA} * omega(ext_k).pow((uint64_t)b);
        return (cb.pow((uint64_t)n) - Fe::one()).inv();
    }
    Fe eval_poly(const Col& h, size_t m, const Fe& x) const {
        Fe out;
        check(ezkl_hip_eval_poly_dev(h->ptr(), m, x.v.data(), out.v.data(), nullptr), "ezkl_hip_eval_poly_dev");
        return out;
    }
    Col slice_copy(const Col& h, size_t off, size_t m) const {
        Col o = alloc(m);
        scale_into(at(h, off), one, o->ptr(), m);
        return o;
    }
    void axpy(const Col& acc, const Fe& s, const Col& h, size_t m) const {       // acc += s * h
        Col t = alloc(m);
        scale_into(h->ptr(), s, t->ptr(), m);
        vec(EZKL_VEC_ADD, acc->ptr(), t->ptr(), acc->ptr(), m);
    }
    // out = (accumulate ? out : 0) + sum_j coeffs[j] * polys[j], one fused pass
    void lincomb(const Col& out, const std::vector<Col>& polys, const std::vector<Fe>& coeffs, size_t m, bool accumulate) const {
        std::vector<const void*> ptrs;
        std::vector<U256> cf;
        for (auto& p_ : polys) ptrs.push_back(p_->ptr());
        for (auto& c_ : coeffs) cf.push_back(c_.v);
        check(ezkl_hip_lincomb_dev(ptrs.data(), cf.data(), (uint32_t)ptrs.size(), out->ptr(), m, accumulate ? 1 : 0, nullptr), "ezkl_hip_lincomb_dev");
    }
    std::vector<Fe> eval_poly_batch(const std::vector<Col>& polys, const std::vector<Fe>& xs, size_t m) const {
        std::vector<const void*> ptrs;
        std::vector<U256> xv;
        for (auto& p_ : polys) ptrs.push_back(p_->ptr());
        for (auto& x_ : xs) xv.push_back(x_.v);
        std::vector<Fe> out(polys.size());
        if (!polys.empty())
            check(ezkl_hip_eval_poly_batch_dev(ptrs.data(), xv.data(), (uint32_t)ptrs.size(), m, out.data(), nullptr), "ezkl_hip_eval_poly_batch_dev");
        return out;
    }
    void sub_low(const Col& h, const std::vector<Fe>& coeffs) const {           // h[i] -= coeffs[i] for the lowest coefficients
        std::vector<U256> v;
        for (auto& c : coeffs) v.push_back(c.v);
        Col t = upload(v);
        vec(EZKL_VEC_SUB, h->ptr(), t->ptr(), h->ptr(), v.size());
    }
    void scale(const Col& h, const Fe& s, size_t m) const { scale_into(h->ptr(), s, h->ptr(), m); }
    void set_rows(const Col& h, size_t start, const std::vector<U256>& rows) const {
        if (!rows.empty()) check(ezkl_hip_memcpy_h2d(at(h, start), rows.data(), rows.size() * 32), "ezkl_hip_memcpy_h2d");
    }
    Fe get_row(const Col& h, size_t i) const {
        Fe out;
        check(ezkl_hip_memcpy_d2h(out.v.data(), at(h, i), 32), "ezkl_hip_memcpy_d2h");
        return out;
    }
    Col omega_powers() const {                                                    // X[i] = omega^i
        Col c = alloc(n);
        fill(c->ptr(), omega(k), n);
        scan(EZKL_VEC_MUL, true, c->ptr(), c->ptr(), n);
        return c;
    }
    // z[0] = z0 (1), z[i+1] = z[i] * prod_j (v_j[i] + beta*delta^(j0+j)*omega^i + gamma) / (v_j[i] + beta*sigma_j[i] + gamma)
    // for one chunk of permutation columns (permutation::prover::commit)
    Col permutation_product(const std::vector<Col>& values, const std::vector<Col>& sigmas, const Fe& beta, const Fe& gamma, uint32_t first_index,
                            const Fe* z0, const Col& omega_col) const {
        const uint32_t m = (uint32_t)values.size();
        std::vector<Col> cols(values);
        cols.insert(cols.end(), sigmas.begin(), sigmas.end());
        cols.push_back(omega_col);
        std::vector<Fe> chal = {beta, gamma};
        const Fe delta{FR_DELTA};
        Fe dp = delta.pow(first_index);
        for (uint32_t j = 0; j < m; j++) { chal.push_back(beta * dp); dp = dp * delta; }
        Program den(k, k), num(k, k);
        Src acc{}, accn{};
        for (uint32_t j = 0; j < m; j++) {
            Src t = den.add(den.add(den.mul(den.challenge(0), den.column(m + j)), den.challenge(1)), den.column(j));
            acc = j == 0 ? t : den.mul(acc, t);
            Src tn = num.add(num.add(num.mul(num.challenge(2 + j), num.column(2 * m)), num.challenge(1)), num.column(j));
            accn = j == 0 ? tn : num.mul(accn, tn);
        }
        Col d_den = alloc(n), d_num = alloc(n);
        den.run(cols, chal, d_den->ptr());
        num.run(cols, chal, d_num->ptr());
        invert(d_den->ptr(), n);
        vec(EZKL_VEC_MUL, d_num->ptr(), d_den->ptr(), d_num->ptr(), n);
        scan(EZKL_VEC_MUL, true, d_num->ptr(), d_num->ptr(), n);
        if (z0) scale(d_num, *z0, n);
        return d_num;
    }
    // all chunks of the permutation argument at once: the denominators of every chunk share ONE batch inversion (see
    // lookup_grand_sums); chunk j starts from the value chunk j-1 reaches on row `usable` (the chaining of permutation::prover::commit).
    // Owner mode: a rank computes the chunks it owns as UNCHAINED running products z'_j (z'_j[0] = 1); z_j = z'_j * prod_{i<j} z'_i[usable],
    // so the only thing the ranks exchange is one field element per chunk.  zs[j] is null for chunks of other ranks.
    std::vector<Col> permutation_products(const std::vector<std::vector<Col>>& values, const std::vector<std::vector<Col>>& sigmas, const Fe& beta,
                                          const Fe& gamma, const Col& omega_col, uint32_t usable) const {
        const size_t nch = values.size();
        std::vector<Col> zs(nch);
        if (!nch) return zs;
        std::vector<size_t> mine;
        for (size_t c = 0; c < nch; c++)
            if (topo.mine(c)) mine.push_back(c);
        Col dens = mine.empty() ? Col() : alloc(mine.size() * n);
        const Fe delta{FR_DELTA};
        std::vector<uint32_t> first(nch, 0);
        for (size_t c = 1; c < nch; c++) first[c] = first[c - 1] + (uint32_t)values[c - 1].size();
        for (size_t q = 0; q < mine.size(); q++) {
            const size_t c = mine[q];
            const uint32_t m = (uint32_t)values[c].size();
            std::vector<Col> cols(values[c]);
            cols.insert(cols.end(), sigmas[c].begin(), sigmas[c].end());
            cols.push_back(omega_col);
            std::vector<Fe> chal = {beta, gamma};
            Fe dp = delta.pow(first[c]);
            for (uint32_t j = 0; j < m; j++) { chal.push_back(beta * dp); dp = dp * delta; }
            Program den(k, k), num(k, k);
            Src acc{}, accn{};
            for (uint32_t j = 0; j < m; j++) {
                Src t = den.add(den.add(den.mul(den.challenge(0), den.column(m + j)), den.challenge(1)), den.column(j));
                acc = j == 0 ? t : den.mul(acc, t);
                Src tn = num.add(num.add(num.mul(num.challenge(2 + j), num.column(2 * m)), num.challenge(1)), num.column(j));
                accn = j == 0 ? tn : num.mul(accn, tn);
            }
            Col d_num = alloc(n);
            den.run(cols, chal, at(dens, q * n));
            num.run(cols, chal, d_num->ptr());
            zs[c] = d_num;
        }
        if (!mine.empty()) invert(dens->ptr(), mine.size() * n);
        std::vector<Fe> lasts((size_t)topo.world * nch, Fe::zero());       // slice r: the unchained z'_c[usable] of rank r's chunks
        for (size_t q = 0; q < mine.size(); q++) {
            const size_t c = mine[q];
            vec(EZKL_VEC_MUL, zs[c]->ptr(), at(dens, q * n), zs[c]->ptr(), n);
            scan(EZKL_VEC_MUL, true, zs[c]->ptr(), zs[c]->ptr(), n);
        }
        for (size_t c : mine)
            if (c + 1 < nch) lasts[(size_t)topo.rank * nch + c] = get_row(zs[c], usable);
        allgather_fe(lasts, nch);
        Fe run = Fe::one();
        for (size_t c = 0; c < nch; c++) {
            if (c && zs[c]) scale(zs[c], run, n);
            if (c + 1 < nch) run = run * lasts[(size_t)topo.owner(c) * nch + c];
        }
        return zs;
    }
    // phi[0] = 0, phi[i+1] = phi[i] + sum_j 1/(f_j[i] + beta) - m[i]/(t[i] + beta)   (mv_lookup::prover::commit_grand_sum)
    Col lookup_grand_sum(const std::vector<Col>& inputs, const Col& table, const Col& m, const Fe& beta) const {
        Col acc = zeros(n), tmp = alloc(n);
        Program shift(k, k);
        shift.add(shift.column(0), shift.challenge(0));
        for (auto& in : inputs) {
            shift.run({in}, {beta}, tmp->ptr());
            invert(tmp->ptr(), n);
            vec(EZKL_VEC_ADD, acc->ptr(), tmp->ptr(), acc->ptr(), n);
        }
        shift.run({table}, {beta}, tmp->ptr());
        invert(tmp->ptr(), n);
        vec(EZKL_VEC_MUL, tmp->ptr(), m->ptr(), tmp->ptr(), n);
        vec(EZKL_VEC_SUB, acc->ptr(), tmp->ptr(), acc->ptr(), n);
        scan(EZKL_VEC_ADD, true, acc->ptr(), acc->ptr(), n);
        return acc;
    }
    // the running sums of ALL lookup arguments at once: the (input + beta) / (table + beta) columns of every argument are laid out in
    // ONE buffer and inverted by ONE batch inversion -- each call ends in a ~380-product dependent Fermat chain (its latency floor,
    // ~0.25 ms whatever the size), so one call for the whole proof instead of one per column (31 calls in a 12-lookup proof: 9 ms)
    std::vector<Col> lookup_grand_sums(const std::vector<std::vector<Col>>& inputs, const std::vector<Col>& tables, const std::vector<Col>& ms,
                                       const Fe& beta) const {
        size_t total = 0;
        for (size_t i = 0; i < tables.size(); i++) total += inputs[i].size() + 1;
        std::vector<Col> out;
        if (!total) return out;
        Col big = alloc(total * n);
        fill(big->ptr(), beta, total * n);                 // (column + beta) for every column: one fill, one addition each -- no program,
        size_t slot = 0;                                   // no host round trip per column
        for (size_t i = 0; i < tables.size(); i++) {
            for (auto& in : inputs[i]) { void* d = at(big, (slot++) * n); vec(EZKL_VEC_ADD, in->ptr(), d, d, n); }
            void* d = at(big, (slot++) * n);
            vec(EZKL_VEC_ADD, tables[i]->ptr(), d, d, n);
        }
        invert(big->ptr(), total * n);
        slot = 0;
        for (size_t i = 0; i < tables.size(); i++) {
            Col acc = zeros(n);
            for (size_t j = 0; j < inputs[i].size(); j++) vec(EZKL_VEC_ADD, acc->ptr(), at(big, (slot++) * n), acc->ptr(), n);
            void* t = at(big, (slot++) * n);
            vec(EZKL_VEC_MUL, t, ms[i]->ptr(), t, n);
            vec(EZKL_VEC_SUB, acc->ptr(), t, acc->ptr(), n);
            scan(EZKL_VEC_ADD, true, acc->ptr(), acc->ptr(), n);
            out.push_back(acc);
        }
        return out;
    }
    // the multiplicity columns of several lookup arguments in one call (three launches for all of them).  `missing`: a resident counter
    // (first u32 of a zeroed column) that collects the inputs absent from their tables over all the arguments; nothing comes back to the
    // host here, the caller checks it once (lookup_check)
    std::vector<Col> lookup_multiplicities(const std::vector<std::vector<Col>>& inputs, const std::vector<Col>& tables, uint32_t usable, const Col& missing) const {
        std::vector<Col> outs;
        if (tables.empty()) return outs;
        std::vector<const void*> in_ptrs, tab_ptrs;
        std::vector<uint32_t> which;
        std::vector<void*> out_ptrs;
        for (size_t l = 0; l < tables.size(); l++) {
            for (auto& c : inputs[l]) { in_ptrs.push_back(c->ptr()); which.push_back((uint32_t)l); }
            tab_ptrs.push_back(tables[l]->ptr());
            outs.push_back(alloc(n));
            out_ptrs.push_back(outs.back()->ptr());
        }
        check(ezkl_hip_lookup_multiplicity_batch_dev(in_ptrs.data(), which.data(), (uint32_t)in_ptrs.size(), tab_ptrs.data(), (uint32_t)tab_ptrs.size(), n, usable,
                                                     out_ptrs.data(), missing->ptr(), nullptr),
              "ezkl_hip_lookup_multiplicity_batch_dev");
        return outs;
    }
    // Owner mode: only the owner of a lookup argument sees its counter, and a rank that threw alone would leave the others waiting in the
    // next collective for ever (the most common ezkl prover failure -- a witness value outside its table -- hung the whole
    // job).  The failure is made COLLECTIVE: every rank contributes its count to one small all_gather and every rank throws the same error.
    void lookup_check(const Col& missing) const {
        const Fe v = get_row(missing, 0);
        uint64_t bad = v.v[0];
        if (topo.owners) {
            std::vector<Fe> all(topo.world, Fe::zero());
            all[topo.rank].v[0] = bad;
            allgather_fe(all, 1);
            bad = 0;
            for (auto& e : all) bad += e.v[0];
        }
        // the reference's mv-lookup prover fails here too (a witness with an input outside the table has no valid proof)
        if (bad != 0) throw Error(EZKL_ERR_INVALID, "lookup input not in table (" + std::to_string((uint32_t)bad) + " rows)");
    }
    // q(X) = p(X) / (X - z) in place (halo2's kate_division)
    void kate_div(const Col& h, const Fe& z, size_t m) const {
        check(ezkl_hip_kate_division_dev(h->ptr(), z.v.data(), h->ptr(), m, nullptr), "ezkl_hip_kate_division_dev");
    }
};

// ------------------------------------------------------------------ keys
struct ProvingKey {
    ConstraintSystem* cs = nullptr;
    std::vector<Col> fixed_values, fixed_polys, fixed_cosets, sigma_values, sigma_polys, sigma_cosets;
    Col omega_col, l0, l_last, l_active, x_coset;
    // which cosets of the extended domain the *_cosets / l0 / l_last / l_active / x_coset columns hold: [coset_first, coset_first +
    // coset_count), coset-major.  One rank / replicated provers: all E of them.  Owner mode: only the cosets this rank sweeps
    // (Backend::key_range) -- at k = 22 the 77 key columns of the 30-column circuit are 39 GB per GPU in full, 1 / world of that by owner
    uint32_t coset_first = 0, coset_count = 0;
    // the degraded mode (EZKL_KEY_COSETS, check_key_fits): fixed_cosets / sigma_cosets hold NULL columns -- the key is values + coefficients
    // (+ l_0 / l_last / l_active / X on the cosets: four columns) and create_proof streams the extended domain one coset at a time
    bool stream = false;
    std::vector<G1> fixed_commitments, sigma_commitments;
    std::vector<uint8_t> selector_bits;      // n_selectors x n/8 bytes, bit-packed rows as in halo2's vk files (zero if the key was made here)
    Fe digest;
};
static Fe vk_digest(const ProvingKey& pk) {
    const ConstraintSystem& cs = *pk.cs;
    // keccak256(keccak256(constraint-system blob) || fixed commitments || permutation commitments): the whole description of
    // the circuit -- columns, gates, lookups, permutation, query order -- is bound into the transcript, as halo2's
    // vk.transcript_repr binds its pinned constraint system
    std::vector<uint8_t> t(cs.blob_hash.begin(), cs.blob_hash.end());
    auto put = [&](const G1& p) {
        U256 x, y;
        p.canonical(x, y);
        uint8_t b[64];
        to_be32(x, b);
        to_be32(y, b + 32);
        t.insert(t.end(), b, b + 64);
    };
    for (auto& p : pk.fixed_commitments) put(p);
    for (auto& p : pk.sigma_commitments) put(p);
    auto h = keccak256(t.data(), t.size());
    return Fe::from_canonical(reduce_fr(from_be32(h.data())));
}
// Before a key is built or loaded (keygen, pk_read, pk_read_file): will it fit, and in which form?  -> true: the key is held STREAMED.
// Resident (the fast form): (fixed + permutation columns) x (values + coefficients + this rank's cosets of the extended domain) + l_0 /
// l_last / l_active / X on those cosets, 32 bytes per element -- at k = 22 / 30 advice columns 61 GB, and a proof holds its witness columns
// (advice, m / phi / compressed inputs, z, h) in the same three forms next to it: 175 of the 288 GB.  k = 23, or a wider circuit at k = 22,
// does not fit ONE GPU that way.  Streamed (the degraded form, one rank only): no column keeps its extended form -- the key is values +
// coefficients, the proof's columns likewise, and the quotient sweep rebuilds coset b of every column from its coefficients when it
// reaches unit b (one n-point transform per column and coset: for witness columns the very transforms the resident form runs earlier, for
// key columns E more per proof).  Peak: 2 n per column + ONE coset of every column, instead of (2 + E) n per column.  The reference can
// do without its resident cosets too: `precompute-coset` is a cargo FEATURE of its halo2 fork (/root/reference/Cargo.toml:218-226,257).
//   EZKL_KEY_COSETS=auto (default): resident if key + estimated witness fit what the device has left, else streamed if that fits, else the
//                   call fails up front with the sizes (EZKL_ERR_NOMEM) -- not somewhere inside a hipMalloc;
//   EZKL_KEY_COSETS=resident: never stream (only the KEY term refuses: it is exact; the witness estimate is a warning on stderr -- the
//                   caller may never prove with this key);     EZKL_KEY_COSETS=recompute: always stream (one rank).
//   EZKL_PROVER_SKIP_FIT_CHECK=1: no check at all (resident unless recompute is asked for); EZKL_PROVER_ASSUME_FREE_GIB=<x> (test hook):
//                   pretend the device has x GiB left.
static bool check_key_fits(const ConstraintSystem& cs, uint32_t coset_count, const char* what, bool one_rank) {
    const char* mode_e = getenv("EZKL_KEY_COSETS");
    const std::string mode = mode_e && *mode_e ? mode_e : "auto";
    invalid(mode != "auto" && mode != "resident" && mode != "recompute", "EZKL_KEY_COSETS must be auto, resident or recompute");
    const uint64_t E = 1ull << (cs.ext_k - cs.k);
    const bool can_stream = one_rank && E > 1;
    invalid(mode == "recompute" && !one_rank, "EZKL_KEY_COSETS=recompute is the one-GPU degraded mode: a sharded prover divides the cosets by the world size instead");
    if (mode == "recompute" && can_stream) return true;
    if (const char* e = getenv("EZKL_PROVER_SKIP_FIT_CHECK")) if (*e && *e != '0') return false;
    const uint64_t n = cs.n, cols = (uint64_t)cs.n_fixed + cs.perm.size();
    const uint64_t key = (cols * (2 + (uint64_t)coset_count) + 4ull * coset_count) * n * 32;
    const double share = (double)coset_count / (double)E;                   // 1 on one rank / replicated; 1 / world by owner
    const uint64_t wcols = (uint64_t)cs.n_advice + 3ull * cs.lookups.size() + cs.n_chunks + 2;
    const uint64_t witness = (uint64_t)((double)(wcols * (2 + E) * n * 32) * share);
    // streamed: two forms per column, ONE coset of every column at a time, h on the whole extended domain + its coefficients
    const uint64_t key_s = (cols * 2 + 4ull * coset_count) * n * 32, witness_s = (wcols * 2 + (cols + wcols) + 2 * E) * n * 32;
    size_t free_b = 0, total_b = 0, pool[4] = {0, 0, 0, 0};
    if (ezkl_hip_mem_info(&free_b, &total_b) != EZKL_OK) return false;      // no device: the first kernel call reports it
    (void)ezkl_hip_pool_stats(pool);
    uint64_t avail = (uint64_t)free_b + pool[2];                            // parked blocks of the column pool are reusable
    if (const char* e = getenv("EZKL_PROVER_ASSUME_FREE_GIB")) avail = (uint64_t)(atof(e) * 1073741824.0);
    if (key + witness <= avail) return false;
    const double G = 1073741824.0;
    if (mode == "auto" && can_stream && key_s + witness_s <= avail) {
        fprintf(stderr, "[ezkl_prover] %s: key %.1f GiB + witness about %.1f GiB do not fit the %.1f GiB available: the key is held STREAMED (values + "
                        "coefficients, %.1f GiB; a proof about %.1f GiB more) and every proof rebuilds the extended domain one coset at a time\n",
                what, key / G, witness / G, avail / G, key_s / G, witness_s / G);
        return true;
    }
    char msg[800];
    snprintf(msg, sizeof msg,
             "%s: the proving key of this circuit needs %.1f GiB resident on the device (%llu key columns x 2^%u rows, %u of %llu cosets of the extended "
             "domain 2^%u) and a proof about %.1f GiB more for its witness columns (streamed, EZKL_KEY_COSETS: %.1f + %.1f GiB%s); %.1f GiB are available "
             "of %.1f GiB. Shard the key over more GPUs (owner mode, ezkl_prover_cs_set_shard_exchange: the coset term divides by the world size), "
             "lower logrows, or set EZKL_PROVER_SKIP_FIT_CHECK=1 to try regardless.",
             what, key / G, (unsigned long long)cols, cs.k, coset_count, (unsigned long long)E, cs.ext_k, witness / G, key_s / G, witness_s / G,
             one_rank ? "" : ", one rank only", avail / G, total_b / G);
    // neither form fits with its proof: only a key that does not fit by ITSELF is refused -- the witness term is an estimate, and the
    // caller may never prove with this key (keygen + pk_write).  The smaller form that still holds the key is taken.
    if (key <= avail) {
        fprintf(stderr, "[ezkl_prover] warning: %s\n", msg);
        return false;
    }
    if (mode == "auto" && can_stream && key_s <= avail) {
        fprintf(stderr, "[ezkl_prover] warning (key held streamed): %s\n", msg);
        return true;
    }
    throw Error(EZKL_ERR_NOMEM, msg);
}
static std::unique_ptr<ProvingKey> keygen(ConstraintSystem& cs, ezkl_bases_t g, const void* const* fixed_values, const uint32_t* copies, size_t n_copies) {
    const uint32_t n = cs.n, k = cs.k;
    Backend be(k, n, g, nullptr, cs.shard);
    {
        uint32_t first = 0, count = 0;
        be.key_range(cs.ext_k, first, count);
        be.stream_cosets = check_key_fits(cs, count, "keygen", !be.topo.owners && !cs.shard.on());
    }
    auto pk = std::make_unique<ProvingKey>();
    pk->cs = &cs;
    pk->stream = be.stream_cosets;
    // EZKL_PROVER_KEYGEN_TIMING=1: stage times on stderr
    const bool timing = getenv("EZKL_PROVER_KEYGEN_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        (void)ezkl_hip_synchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[ezkl_prover] keygen %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    for (uint32_t c = 0; c < cs.n_fixed; c++) {
        pk->fixed_values.push_back(be.upload(fixed_values[c], n));
        pk->fixed_polys.push_back(be.lagrange_to_coeff(pk->fixed_values.back()));
        pk->fixed_cosets.push_back(pk->stream ? Col() : be.key_cosets(pk->fixed_polys.back(), cs.ext_k));
    }
    lap("fixed columns");
    // permutation: cycle structure over (colpos, row) cells numbered c * n + r; `nxt` is the cycle successor, `root` a
    // cycle label, `size` the cycle length (halo2 permutation::keygen::Assembly::copy)
    const size_t m = cs.perm.size(), cells = m * n;
    std::vector<uint32_t> nxt(cells), root(cells), size(cells, 1);
    for (size_t i = 0; i < cells; i++) nxt[i] = root[i] = (uint32_t)i;
    for (size_t i = 0; i < n_copies; i++) {
        const uint32_t* cp = copies + 4 * i;
        invalid(cp[0] >= m || cp[2] >= m || cp[1] >= n || cp[3] >= n, "copy constraint out of range");
        uint32_t a = cp[0] * n + cp[1], b = cp[2] * n + cp[3];
        if (root[a] == root[b]) continue;
        if (size[root[a]] < size[root[b]]) std::swap(a, b);
        const uint32_t ra = root[a], rb = root[b];
        size[ra] += size[rb];
        uint32_t cur = b;
        do {                                     // relabel the smaller cycle
            root[cur] = ra;
            cur = nxt[cur];
        } while (cur != b);
        std::swap(nxt[a], nxt[b]);
    }
    lap("copy cycles (host)");
    // sigma[c][r] = delta^c' * omega^r' for (c', r') = nxt[(c, r)]: gathered on the device (ezkl_hip_permutation_sigma_dev) from the
    // resident omega^r column and the m powers of delta; only the successor map travels (4 B per cell)
    pk->omega_col = be.omega_powers();
    {
        std::vector<U256> dpv(m ? m : 1, U256{0, 0, 0, 0});
        const Fe delta{FR_DELTA};
        Fe dp = Fe::one();
        for (size_t c = 0; c < m; c++) {
            dpv[c] = dp.v;
            dp = dp * delta;
        }
        const Col dpc = be.upload(dpv);
        const Col next = be.alloc((n + 7) / 8);                      // n u32 successors
        lap("delta powers");
        for (size_t c = 0; c < m; c++) {
            check(ezkl_hip_memcpy_h2d(next->ptr(), nxt.data() + c * n, (size_t)n * 4), "ezkl_hip_memcpy_h2d");
            Col sig = be.alloc(n);
            check(ezkl_hip_permutation_sigma_dev(next->ptr(), pk->omega_col->ptr(), dpc->ptr(), (uint32_t)m, k, sig->ptr(), nullptr), "ezkl_hip_permutation_sigma_dev");
            pk->sigma_values.push_back(sig);
            pk->sigma_polys.push_back(be.lagrange_to_coeff(pk->sigma_values.back()));
            pk->sigma_cosets.push_back(pk->stream ? Col() : be.key_cosets(pk->sigma_polys.back(), cs.ext_k));
        }
    }
    lap("sigma gather + forms");
    // l0, l_last, l_active_row on the extended coset
    auto lag = [&](uint32_t lo, uint32_t hi) { return be.key_cosets(be.lagrange_to_coeff(be.indicator(lo, hi)), cs.ext_k); };
    pk->l0 = lag(0, 1);
    pk->l_last = lag(cs.usable, cs.usable + 1);
    pk->l_active = lag(0, cs.usable);
    // the identity column X on the extended coset (from coefficients [0, 1, 0, ...])
    pk->x_coset = be.key_cosets(be.indicator(1, 2), cs.ext_k);
    be.key_range(cs.ext_k, pk->coset_first, pk->coset_count);
    lap("l0 / l_last / l_active / X");
    pk->fixed_commitments = be.commit(pk->fixed_polys);
    pk->sigma_commitments = be.commit(pk->sigma_polys);
    pk->digest = vk_digest(*pk);
    lap("commitments");
    return pk;
}

// ------------------------------------------------------------------ key files (halo2 ProvingKey::{write, read}, SerdeFormat::RawBytes)
// The layout of the reference's vk.key / pk.key (/root/reference/src/pfsys/mod.rs:593-683; verified on tests/assets in
// SURVEY.md §8(c) item 3):  VK = [3, k, compress_selectors] | u32 LE #fixed | #fixed x G1 | #perm x G1 | selectors (none
// here: selectors are plain fixed columns);  PK = VK | poly l0 | poly l_last | poly l_active_row | vec fixed_values |
// vec fixed_polys | vec fixed_cosets | vec permutations | vec perm_polys | vec perm_cosets, with
// poly = u32 BE len | len x 32 B and vec = u32 BE count | count x u32 BE len | count x poly.  Field and curve bytes are
// the resident Montgomery bytes, copied unchanged in both directions.
static void put_be32(std::vector<uint8_t>& o, uint32_t v) {
    for (int i = 3; i >= 0; i--) o.push_back((uint8_t)(v >> (8 * i)));
}
static std::vector<uint8_t> pk_write(const ProvingKey& pk) {
    const ConstraintSystem& cs = *pk.cs;
    Backend be(cs.k, cs.n, nullptr, nullptr);
    std::vector<uint8_t> o = {3, (uint8_t)cs.k, 1};
    const uint32_t nf = cs.n_fixed;
    for (int i = 0; i < 4; i++) o.push_back((uint8_t)(nf >> (8 * i)));
    auto put_points = [&](const std::vector<G1>& v) {
        const uint8_t* b = reinterpret_cast<const uint8_t*>(v.data());
        o.insert(o.end(), b, b + 64 * v.size());
    };
    put_points(pk.fixed_commitments);
    put_points(pk.sigma_commitments);
    {
        const size_t sel_bytes = (size_t)cs.n_selectors * ((cs.n + 7) / 8);
        if (pk.selector_bits.size() == sel_bytes) o.insert(o.end(), pk.selector_bits.begin(), pk.selector_bits.end());
        else o.insert(o.end(), sel_bytes, 0);
    }
    auto put_poly = [&](const Col& c, size_t m) {
        std::vector<U256> v = be.download(c, m);
        put_be32(o, (uint32_t)m);
        const uint8_t* b = reinterpret_cast<const uint8_t*>(v.data());
        o.insert(o.end(), b, b + 32 * m);
    };
    auto put_vec = [&](const std::vector<Col>& cols, size_t m) {
        put_be32(o, (uint32_t)cols.size());
        for (size_t i = 0; i < cols.size(); i++) put_be32(o, (uint32_t)m);
        for (auto& c : cols) put_poly(c, m);
    };
    const size_t n = cs.n, ne = (size_t)1 << cs.ext_k;
    // the resident extended columns are coset-major; the file holds halo2's natural order of the extended domain
    auto nat = [&](const Col& c) { return be.cosets_reorder(c, cs.ext_k, true); };
    auto put_ext_vec = [&](const std::vector<Col>& cols) {
        put_be32(o, (uint32_t)cols.size());
        for (size_t i = 0; i < cols.size(); i++) put_be32(o, (uint32_t)ne);
        for (auto& c : cols) put_poly(nat(c), ne);
    };
    if (pk.coset_count == (1u << (cs.ext_k - cs.k)) && !pk.stream) {
        put_poly(nat(pk.l0), ne); put_poly(nat(pk.l_last), ne); put_poly(nat(pk.l_active), ne);
        put_vec(pk.fixed_values, n); put_vec(pk.fixed_polys, n); put_ext_vec(pk.fixed_cosets);
        put_vec(pk.sigma_values, n); put_vec(pk.sigma_polys, n); put_ext_vec(pk.sigma_cosets);
    } else {
        // a key held by owner (a range of cosets) or streamed (no extended columns at all): the file still gets the complete extended columns, recomputed from the coefficient forms
        // ... ONE column at a time (coefficients -> extended -> natural order -> host -> freed): materialising every extended column first
        // was 35-39 GB of HBM at k = 22 on the rank whose key had deliberately been cut to 1 / world
        auto put_ext_from_polys = [&](const std::vector<Col>& polys) {
            put_be32(o, (uint32_t)polys.size());
            for (size_t i = 0; i < polys.size(); i++) put_be32(o, (uint32_t)ne);
            for (auto& p_ : polys) {
                Col ext = be.coeff_to_extended(p_, cs.ext_k);
                put_poly(nat(ext), ne);
            }
        };
        auto lagf = [&](uint32_t lo, uint32_t hi) { return be.coeff_to_extended(be.lagrange_to_coeff(be.indicator(lo, hi)), cs.ext_k); };
        put_poly(nat(lagf(0, 1)), ne); put_poly(nat(lagf(cs.usable, cs.usable + 1)), ne); put_poly(nat(lagf(0, cs.usable)), ne);
        put_vec(pk.fixed_values, n); put_vec(pk.fixed_polys, n); put_ext_from_polys(pk.fixed_polys);
        put_vec(pk.sigma_values, n); put_vec(pk.sigma_polys, n); put_ext_from_polys(pk.sigma_polys);
    }
    return o;
}
static std::unique_ptr<ProvingKey> pk_read(ConstraintSystem& cs, const uint8_t* buf, size_t len) {
    Backend be(cs.k, cs.n, nullptr, nullptr);
    const bool stream = check_key_fits(cs, 1u << (cs.ext_k - cs.k), "pk_read", !cs.shard.on());   // the file's complete extended columns
    size_t off = 0;
    auto need = [&](size_t m) { invalid(off + m > len, "proving key truncated"); };
    need(7);
    invalid(buf[0] != 3, "unsupported key version");
    invalid(buf[1] != cs.k, "key was made for another k");
    uint32_t nf = 0;
    for (int i = 0; i < 4; i++) nf |= (uint32_t)buf[3 + i] << (8 * i);
    invalid(nf != cs.n_fixed, "key has another number of fixed columns");
    off = 7;
    auto pk = std::make_unique<ProvingKey>();
    pk->cs = &cs;
    auto get_points = [&](std::vector<G1>& v, size_t m) {
        need(64 * m);
        v.resize(m);
        if (m) std::memcpy(v.data(), buf + off, 64 * m);
        off += 64 * m;
    };
    get_points(pk->fixed_commitments, cs.n_fixed);
    get_points(pk->sigma_commitments, cs.perm.size());
    {
        // halo2 does not store the selector count: it re-runs configure (src/pfsys/mod.rs:627); here the constraint system carries it
        const size_t sel_bytes = (size_t)cs.n_selectors * ((cs.n + 7) / 8);
        need(sel_bytes);
        pk->selector_bits.assign(buf + off, buf + off + sel_bytes);
        off += sel_bytes;
    }
    auto be32 = [&]() {
        need(4);
        uint32_t v = ((uint32_t)buf[off] << 24) | ((uint32_t)buf[off + 1] << 16) | ((uint32_t)buf[off + 2] << 8) | buf[off + 3];
        off += 4;
        return v;
    };
    auto get_poly = [&](size_t m) {
        invalid(be32() != m, "polynomial of unexpected length in the key");
        need(32 * m);
        for (size_t i = 0; i < m; i++) {                               // every element must be a canonical residue
            U256 e;
            std::memcpy(e.data(), buf + off + 32 * i, 32);
            invalid(cmp(e, FR.p) >= 0, "non-canonical field element in the key");
        }
        Col c = be.upload(buf + off, m);
        off += 32 * m;
        return c;
    };
    auto get_vec = [&](std::vector<Col>& cols, size_t count, size_t m) {
        invalid(be32() != count, "vector of unexpected length in the key");
        for (size_t i = 0; i < count; i++) invalid(be32() != m, "polynomial of unexpected length in the key");
        for (size_t i = 0; i < count; i++) cols.push_back(get_poly(m));
    };
    const size_t n = cs.n, ne = (size_t)1 << cs.ext_k;
    // extended columns: natural order in the file, coset-major in HBM
    auto cm = [&](const Col& c) { return be.cosets_reorder(c, cs.ext_k, false); };
    pk->l0 = cm(get_poly(ne)); pk->l_last = cm(get_poly(ne)); pk->l_active = cm(get_poly(ne));
    // streamed key: the extended sections of the file are checked for their shape and passed over, never uploaded
    auto skip_vec = [&](std::vector<Col>& cols, size_t count, size_t m) {
        invalid(be32() != count, "vector of unexpected length in the key");
        for (size_t i = 0; i < count; i++) invalid(be32() != m, "polynomial of unexpected length in the key");
        for (size_t i = 0; i < count; i++) {
            invalid(be32() != m, "polynomial of unexpected length in the key");
            need(32 * m);
            off += 32 * m;
            cols.push_back(Col());
        }
    };
    pk->stream = stream;
    get_vec(pk->fixed_values, cs.n_fixed, n); get_vec(pk->fixed_polys, cs.n_fixed, n);
    if (stream) skip_vec(pk->fixed_cosets, cs.n_fixed, ne); else get_vec(pk->fixed_cosets, cs.n_fixed, ne);
    get_vec(pk->sigma_values, cs.perm.size(), n); get_vec(pk->sigma_polys, cs.perm.size(), n);
    if (stream) skip_vec(pk->sigma_cosets, cs.perm.size(), ne); else get_vec(pk->sigma_cosets, cs.perm.size(), ne);
    for (auto& c : pk->fixed_cosets) if (c) c = cm(c);
    for (auto& c : pk->sigma_cosets) if (c) c = cm(c);
    invalid(off != len, "trailing bytes in the proving key");
    // derived columns that the file does not hold
    pk->omega_col = be.omega_powers();
    pk->x_coset = be.coeff_to_extended(be.indicator(1, 2), cs.ext_k);
    pk->coset_first = 0;
    pk->coset_count = 1u << (cs.ext_k - cs.k);               // the file's complete extended columns
    pk->digest = vk_digest(*pk);
    return pk;
}

// `load_pk` for a one-shot prover (/root/reference/src/pfsys/mod.rs:615-636, called from execute::prove): the key file is mapped, only
// what cannot be recomputed faster than it can be read goes over PCIe -- the n-row `fixed_values` and `permutations` sections
// (32 B x n per column); the coefficient forms, the extended cosets (the bulk of the file: 2^(ext_k - k) x larger) and l0 / l_last /
// l_active are recomputed on the device with the kernels keygen uses (an iNTT + a coset NTT per column: milliseconds, against a read
// of GiBs at k = 20).  The section headers are still checked against the constraint system; the commitments come from the file.
static std::unique_ptr<ProvingKey> pk_read_file(ConstraintSystem& cs, const char* path) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) throw Error(EZKL_ERR_INVALID, std::string("cannot open proving key ") + path);
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); throw Error(EZKL_ERR_INVALID, "cannot stat proving key"); }
    const size_t len = (size_t)sb.st_size;
    struct Close { int fd; ~Close() { close(fd); } } close_fd{fd};
    void* map = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);              // the headers are parsed through the mapping
    if (map == MAP_FAILED) throw Error(EZKL_ERR_INVALID, "cannot map proving key");
    struct Unmap { void* p; size_t l; ~Unmap() { munmap(p, l); } } unmap{map, len};
    (void)madvise(map, len, MADV_SEQUENTIAL);
    const uint8_t* buf = (const uint8_t*)map;
    Backend be(cs.k, cs.n, nullptr, nullptr, cs.shard);       // owner mode: only the cosets this rank sweeps are computed and kept
    {
        uint32_t first = 0, count = 0;
        be.key_range(cs.ext_k, first, count);
        be.stream_cosets = check_key_fits(cs, count, "pk_read_file", !be.topo.owners && !cs.shard.on());
    }
    size_t off = 0;
    auto need = [&](size_t m) { invalid(off + m > len, "proving key truncated"); };
    need(7);
    invalid(buf[0] != 3, "unsupported key version");
    invalid(buf[1] != cs.k, "key was made for another k");
    uint32_t nf = 0;
    for (int i = 0; i < 4; i++) nf |= (uint32_t)buf[3 + i] << (8 * i);
    invalid(nf != cs.n_fixed, "key has another number of fixed columns");
    off = 7;
    auto pk = std::make_unique<ProvingKey>();
    pk->stream = be.stream_cosets;
    pk->cs = &cs;
    const size_t n = cs.n, ne = (size_t)1 << cs.ext_k, np = cs.perm.size();
    need(64 * (nf + np));
    pk->fixed_commitments.resize(nf);
    pk->sigma_commitments.resize(np);
    if (nf) std::memcpy(pk->fixed_commitments.data(), buf + off, 64 * (size_t)nf);
    off += 64 * (size_t)nf;
    if (np) std::memcpy(pk->sigma_commitments.data(), buf + off, 64 * np);
    off += 64 * np;
    const size_t sel_bytes = (size_t)cs.n_selectors * ((cs.n + 7) / 8);
    need(sel_bytes);
    pk->selector_bits.assign(buf + off, buf + off + sel_bytes);
    off += sel_bytes;
    auto be32 = [&]() {
        need(4);
        uint32_t v = ((uint32_t)buf[off] << 24) | ((uint32_t)buf[off + 1] << 16) | ((uint32_t)buf[off + 2] << 8) | buf[off + 3];
        off += 4;
        return v;
    };
    auto skip_poly = [&](size_t m) {
        invalid(be32() != m, "polynomial of unexpected length in the key");
        need(32 * m);
        off += 32 * m;
    };
    auto vec_header = [&](size_t count, size_t m) {
        invalid(be32() != count, "vector of unexpected length in the key");
        for (size_t i = 0; i < count; i++) invalid(be32() != m, "polynomial of unexpected length in the key");
    };
    // the n-row sections are only LOCATED here; they travel afterwards, all at once (load_sections below)
    struct Section { size_t off; Col dst; };
    std::vector<Section> sections;
    auto load_values = [&](std::vector<Col>& cols, size_t count) {
        vec_header(count, n);
        for (size_t i = 0; i < count; i++) {
            invalid(be32() != n, "polynomial of unexpected length in the key");
            need(32 * n);
            cols.push_back(be.alloc(n));
            sections.push_back(Section{off, cols.back()});
            off += 32 * n;
        }
    };
    auto skip_vec = [&](size_t count, size_t m) {
        vec_header(count, m);
        for (size_t i = 0; i < count; i++) skip_poly(m);
    };
    skip_poly(ne); skip_poly(ne); skip_poly(ne);                    // l0, l_last, l_active_row: recomputed
    load_values(pk->fixed_values, nf);
    skip_vec(nf, n); skip_vec(nf, ne);
    load_values(pk->sigma_values, np);
    skip_vec(np, n); skip_vec(np, ne);
    invalid(off != len, "trailing bytes in the proving key");
    // File -> HBM as a pipeline: a few reader threads pread their sections into page-locked buffers (one copy out of the page cache,
    // no page fault per 4 KiB as through the mapping) and check them (canonical residues: the top limb decides all but 2^-60 of the
    // cases), this thread uploads each buffer as it fills (PCIe at the pinned rate).  Measured on the k = 20 MLP key (34 sections of
    // 32 MiB inside a 7.8 GB file): NOTEBOOK.md §4.7.
    {
        const size_t bytes = 32 * n, N = sections.size();
        for (auto& sec : sections) (void)posix_fadvise(fd, (off_t)sec.off, (off_t)bytes, POSIX_FADV_WILLNEED);   // a key not in the page cache: read ahead
        // reader threads: 6 by default (k = 20: 34 sections of 32 MiB in 0.17 s); EZKL_PK_READ_THREADS overrides (k = 22: 77 sections of 128 MiB)
        size_t want_threads = 6;
        if (const char* e = getenv("EZKL_PK_READ_THREADS")) { const long v = atol(e); if (v >= 1 && v <= 64) want_threads = (size_t)v; }
        const size_t T = std::min<size_t>({N, want_threads, std::max(1u, std::thread::hardware_concurrency())});
        std::vector<void*> pinned(T, nullptr);
        struct Free { std::vector<void*>& v; ~Free() { for (void* q : v) if (q) (void)ezkl_hip_host_free(q); } } free_pinned{pinned};
        for (auto& q : pinned) check(ezkl_hip_host_malloc(&q, bytes), "ezkl_hip_host_malloc");
        std::mutex mu;
        std::condition_variable cv;
        std::vector<long> holds(T, -1);                              // section held by buffer t (filled, not yet uploaded)
        std::string failure;
        bool stop = false;
        auto reader = [&](size_t t) {
            for (size_t i = t; i < N; i += T) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return holds[t] < 0 || stop; });
                    if (stop) return;
                }
                std::string err;
                uint8_t* q = (uint8_t*)pinned[t];
                for (size_t got = 0; got < bytes && err.empty();) {
                    const ssize_t m = pread(fd, q + got, bytes - got, (off_t)(sections[i].off + got));
                    if (m <= 0) err = "cannot read the proving key";
                    else got += (size_t)m;
                }
                const uint64_t* e = (const uint64_t*)q;
                for (size_t r = 0; r < n && err.empty(); r++)
                    if (e[4 * r + 3] >= FR.p[3]) {
                        U256 v;
                        std::memcpy(v.data(), e + 4 * r, 32);
                        if (cmp(v, FR.p) >= 0) err = "non-canonical field element in the key";
                    }
                std::lock_guard<std::mutex> lk(mu);
                if (!err.empty()) { failure = err; stop = true; }
                else holds[t] = (long)i;
                cv.notify_all();
                if (stop) return;
            }
        };
        std::vector<std::thread> th;
        for (size_t t = 0; t < T; t++) th.emplace_back(reader, t);
        struct Join {
            std::vector<std::thread>& th; std::mutex& mu; std::condition_variable& cv; bool& stop;
            ~Join() {
                { std::lock_guard<std::mutex> lk(mu); stop = true; }
                cv.notify_all();
                for (auto& x : th) if (x.joinable()) x.join();
            }
        } join{th, mu, cv, stop};
        for (size_t i = 0; i < N; i++) {
            const size_t t = i % T;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return holds[t] == (long)i || stop; });
                if (stop) throw Error(EZKL_ERR_INVALID, failure.empty() ? "proving key read failed" : failure);
            }
            check(ezkl_hip_memcpy_h2d(sections[i].dst->ptr(), pinned[t], bytes), "ezkl_hip_memcpy_h2d");
            {
                std::lock_guard<std::mutex> lk(mu);
                holds[t] = -1;
            }
            cv.notify_all();
        }
    }
    // (the transforms run AFTER the uploads: queuing each column's behind its upload, under the readers, was measured slower on the same
    // box and artefacts -- key load 0.184 against 0.160 s, 4 runs each, profiles/r03ai_cold_ab.log: the uploads then wait for the
    // transforms queued before them on the library stream)
    {
        for (auto& v : pk->fixed_values) {
            pk->fixed_polys.push_back(be.lagrange_to_coeff(v));
            pk->fixed_cosets.push_back(pk->stream ? Col() : be.key_cosets(pk->fixed_polys.back(), cs.ext_k));
        }
        for (auto& v : pk->sigma_values) {
            pk->sigma_polys.push_back(be.lagrange_to_coeff(v));
            pk->sigma_cosets.push_back(pk->stream ? Col() : be.key_cosets(pk->sigma_polys.back(), cs.ext_k));
        }
    }
    auto lag = [&](uint32_t lo, uint32_t hi) { return be.key_cosets(be.lagrange_to_coeff(be.indicator(lo, hi)), cs.ext_k); };
    pk->l0 = lag(0, 1);
    pk->l_last = lag(cs.usable, cs.usable + 1);
    pk->l_active = lag(0, cs.usable);
    pk->omega_col = be.omega_powers();
    pk->x_coset = be.key_cosets(be.indicator(1, 2), cs.ext_k);
    be.key_range(cs.ext_k, pk->coset_first, pk->coset_count);
    pk->digest = vk_digest(*pk);
    return pk;
}

// commitments of the fixed / permutation polynomials under another SRS (a key file made with the public SRS, proved here under a
// test SRS): the resident polynomials are committed again and the digest follows
static void pk_recommit(ProvingKey& pk, ezkl_bases_t g) {
    ConstraintSystem& cs = *pk.cs;
    Backend be(cs.k, cs.n, g, nullptr, cs.shard);
    pk.fixed_commitments = be.commit(pk.fixed_polys);
    pk.sigma_commitments = be.commit(pk.sigma_polys);
    pk.digest = vk_digest(pk);
}

// ------------------------------------------------------------------ randomness
// Blinding rows and the vanishing argument's random polynomial.  With a caller-supplied generator (ezkl_rng_fn) the
// elements come from the callback.  Otherwise they are ChaCha20 output under a 256-bit key -- OS entropy, or derived
// from the seed (the reference's det-prove feature, /root/reference/src/pfsys/mod.rs:436-439) -- sampled uniformly on
// [0, r) exactly as ezkl_hip_chacha20_fr_dev does (include/ezkl_hip.h): request number `calls` is stream `calls`, so a
// short vector made on the host and a whole column expanded on the device are the same function of (key, calls, i).
static void chacha20_block(const uint32_t key[8], uint64_t counter, uint64_t stream, uint32_t out[16]) {
    const uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                            (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t x[16];
    std::memcpy(x, s, sizeof s);
    auto rotl = [](uint32_t v, int n) { return (v << n) | (v >> (32 - n)); };
    auto qr = [&](int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16);
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);
        x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
    };
    for (int r = 0; r < 10; r++) {
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}
struct Rng {
    ezkl_rng_fn fn;
    void* user;
    uint32_t key[8];
    uint64_t calls = 0;
    Rng(ezkl_rng_fn f, void* u, uint64_t seed) : fn(f), user(u) {
        if (seed == 0) {
            std::random_device rd;               // /dev/urandom
            for (auto& w : key) w = rd();
        } else {
            uint8_t buf[26] = "ezkl_hip det-prove";
            for (int i = 0; i < 8; i++) buf[18 + i] = (uint8_t)(seed >> (8 * i));
            auto h = keccak256(buf, sizeof buf);
            std::memcpy(key, h.data(), 32);
        }
    }
    std::vector<U256> vec(size_t m) {
        std::vector<U256> out(m, U256{0, 0, 0, 0});
        if (m == 0) return out;
        if (fn) {
            fn(user, out.data(), m);
            for (auto& e : out) invalid(cmp(e, FR.p) >= 0, "rng callback returned a non-canonical residue");
            return out;
        }
        const uint64_t stream = calls++;
        for (size_t i = 0; i < m; i++) {
            bool done = false;
            for (uint32_t b = 0; b < 16 && !done; b++) {
                uint32_t blk[16];
                chacha20_block(key, (uint64_t)i * 16 + b, stream, blk);
                for (int h = 0; h < 2 && !done; h++) {
                    U256 cand;
                    for (int q = 0; q < 4; q++) cand[q] = (uint64_t)blk[8 * h + 2 * q] | ((uint64_t)blk[8 * h + 2 * q + 1] << 32);
                    cand[3] &= 0x3fffffffffffffffull;
                    if (cmp(cand, FR.p) < 0) { out[i] = cand; done = true; }
                }
            }
        }
        return out;
    }
    // a whole column of randomness, made where it lives
    Col column(const Backend& be, size_t m) {
        if (fn) return be.upload(vec(m));
        Col c = be.alloc(m);
        check(ezkl_hip_chacha20_fr_dev(key, calls++, 0, c->ptr(), m, nullptr), "ezkl_hip_chacha20_fr_dev");
        return c;
    }
};

// ------------------------------------------------------------------ SHPLONK (BDFG20) multi-point opening
struct OpenQuery {
    std::vector<uint32_t> key;       // polynomial identity (kind, index)
    Col poly;                        // null on ranks that do not own the polynomial (owner mode)
    Fe point, eval;
    bool mine = true;                // this rank carries the polynomial through the opening (every polynomial has exactly one such rank)
};
struct PolyEvals {
    Col poly;
    bool mine;
    std::map<U256, std::pair<Fe, Fe>> ev;      // canonical point -> (point, eval)
};
static bool u256_less(const U256& a, const U256& b) { return cmp(a, b) < 0; }
// coefficients (low first) of the polynomial of degree < len(points) through (points, values)
static std::vector<Fe> interpolate(const std::vector<Fe>& pts, const std::vector<Fe>& vals) {
    const size_t m = pts.size();
    std::vector<Fe> coeffs(m, Fe::zero()), den(m, Fe::one()), pre(m + 1, Fe::one());
    std::vector<std::vector<Fe>> nums(m);
    for (size_t i = 0; i < m; i++) {
        std::vector<Fe> num = {Fe::one()};
        for (size_t j = 0; j < m; j++) {
            if (j == i) continue;
            std::vector<Fe> nn(num.size() + 1, Fe::zero());
            for (size_t t = 0; t < num.size(); t++) {            // num * (X - pts[j])
                nn[t + 1] = nn[t + 1] + num[t];
                nn[t] = nn[t] - pts[j] * num[t];
            }
            num = nn;
            den[i] = den[i] * (pts[i] - pts[j]);
        }
        nums[i] = num;
        pre[i + 1] = pre[i] * den[i];
    }
    Fe inv_all = pre[m].inv();                                   // one field inversion (254 squarings on the host) for the m denominators
    for (size_t i = m; i-- > 0;) {
        const Fe s = vals[i] * (inv_all * pre[i]);
        inv_all = inv_all * den[i];
        for (size_t t = 0; t < nums[i].size(); t++) coeffs[t] = coeffs[t] + s * nums[i][t];
    }
    return coeffs;
}
static Fe eval_small(const std::vector<Fe>& c, const Fe& x) {
    Fe acc = Fe::zero();
    for (size_t i = c.size(); i-- > 0;) acc = acc * x + c[i];
    return acc;
}
// The opening is LINEAR in the polynomials: q_S = sum_i ys^i p_i, h = sum_S v^S (q_S - r_S) / Z_S, L = sum_S c_S (q_S - r_S(u)) - c h.
// A sharded prover in owner mode therefore never moves a polynomial: every rank forms the same expressions over the polynomials it
// owns (with the global coefficients ys^i, v^S, c_S and the partial evaluations of ITS polynomials: each partial q_S - r_S still
// vanishes on S, so the divisions stay exact), commits its partial h and L with the whole base set, and the two folds add the
// points.  With one rank (or replicated columns) every polynomial is `mine` and this is the plain prover.
static void shplonk_prove(Backend& be, EvmTranscript& T, const std::vector<OpenQuery>& qs, uint32_t n) {
    // group queries by polynomial (first appearance), then polynomials by their point set (first appearance)
    std::vector<PolyEvals> polys;
    std::map<std::vector<uint32_t>, size_t> by_key;
    for (auto& q : qs) {
        auto it = by_key.find(q.key);
        if (it == by_key.end()) {
            it = by_key.emplace(q.key, polys.size()).first;
            polys.push_back(PolyEvals{q.poly, q.mine, {}});
        }
        polys[it->second].ev[q.point.canonical()] = {q.point, q.eval};
    }
    struct Group {
        std::vector<U256> pts;                 // sorted canonical points
        std::vector<Fe> pts_fe;
        std::vector<size_t> members;
    };
    std::vector<Group> groups;
    for (size_t i = 0; i < polys.size(); i++) {
        std::vector<U256> pts;
        for (auto& e : polys[i].ev) pts.push_back(e.first);      // std::map<U256>: lexicographic on LE limbs, re-sort numerically
        std::sort(pts.begin(), pts.end(), u256_less);
        size_t gi = 0;
        for (; gi < groups.size(); gi++)
            if (groups[gi].pts == pts) break;
        if (gi == groups.size()) {
            Group gnew;
            gnew.pts = pts;
            for (auto& p : pts) gnew.pts_fe.push_back(polys[i].ev[p].first);
            groups.push_back(gnew);
        }
        groups[gi].members.push_back(i);
    }
    const Fe ys = T.squeeze_challenge();
    std::vector<U256> all_pts;
    for (auto& gr : groups)
        for (auto& p : gr.pts) all_pts.push_back(p);
    std::sort(all_pts.begin(), all_pts.end(), u256_less);
    all_pts.erase(std::unique(all_pts.begin(), all_pts.end()), all_pts.end());
    std::map<U256, Fe> pt_fe;
    for (auto& gr : groups)
        for (size_t i = 0; i < gr.pts.size(); i++) pt_fe[gr.pts[i]] = gr.pts_fe[i];
    struct Combo {
        Col q;                 // null: this rank owns no member of the group
        std::vector<Fe> r;
    };
    std::vector<Combo> combos;
    for (auto& gr : groups) {
        std::vector<Fe> evs(gr.pts.size(), Fe::zero()), cf;
        std::vector<Col> members;
        Fe pw = Fe::one();
        for (size_t mi : gr.members) {
            if (polys[mi].mine) {
                members.push_back(polys[mi].poly);
                cf.push_back(pw);
                for (size_t i = 0; i < gr.pts.size(); i++) evs[i] = evs[i] + pw * polys[mi].ev[gr.pts[i]].second;
            }
            pw = pw * ys;
        }
        Col q;
        if (!members.empty()) {
            q = be.alloc(n);
            be.lincomb(q, members, cf, n, false);
        }
        combos.push_back(Combo{q, interpolate(gr.pts_fe, evs)});
    }
    const Fe v = T.squeeze_challenge();
    Col h = be.alloc(n);
    Fe pw = Fe::one();
    {
        std::vector<Col> ts;
        std::vector<Fe> cf;
        for (size_t gi = 0; gi < groups.size(); gi++) {
            if (combos[gi].q) {
                Col t = be.clone(combos[gi].q);
                be.sub_low(t, combos[gi].r);
                for (auto& z : groups[gi].pts_fe) be.kate_div(t, z, n);
                ts.push_back(t);
                cf.push_back(pw);
            }
            pw = pw * v;
        }
        if (ts.empty()) be.fill(h->ptr(), Fe::zero(), n);
        else be.lincomb(h, ts, cf, n, false);
    }
    T.write_point(be.commit_sum(be.g, h));
    const Fe u = T.squeeze_challenge();
    Fe zt_u = Fe::one();
    for (auto& z : all_pts) zt_u = zt_u * (u - pt_fe[z]);
    Col L = be.alloc(n);
    pw = Fe::one();
    Fe const_term = Fe::zero();
    {
        std::vector<Col> terms;
        std::vector<Fe> cf;
        Fe norm = Fe::one();                               // halo2 normalises by the first set's coefficient: 1 / Z_{T \ S_0}(u)
        for (size_t gi = 0; gi < groups.size(); gi++) {
            Fe zdiff = Fe::one();
            for (auto& z : all_pts)
                if (!std::binary_search(groups[gi].pts.begin(), groups[gi].pts.end(), z, u256_less)) zdiff = zdiff * (u - pt_fe[z]);
            if (gi == 0) norm = zdiff.inv();
            const Fe c = pw * zdiff * norm;
            if (combos[gi].q) {
                terms.push_back(combos[gi].q);
                cf.push_back(c);
                const_term = const_term + c * eval_small(combos[gi].r, u);
            }
            pw = pw * v;
        }
        terms.push_back(h);
        cf.push_back(-(zt_u * norm));                     // Z_T(u) / Z_{T \ S_0}(u) = Z_{S_0}(u)
        be.lincomb(L, terms, cf, n, false);
    }
    be.sub_low(L, {const_term});
    be.kate_div(L, u, n);
    T.write_point(be.commit_sum(be.g, L));
}

// ------------------------------------------------------------------ the numerator of h(X)
struct Quotient {
    std::vector<Program> progs;    // run one after the other on the same output: program p continues the Horner chain from PreviousValue
    std::vector<Col> cols;         // coset-major extended columns (null: a witness column this rank does not own)
    std::vector<int> owner;        // per slot: the rank holding the column, -1 = resident on every rank (key columns, replicated provers)
    std::vector<uint8_t> key;      // per slot: a key column (holds cosets [pk.coset_first, +pk.coset_count) only)
    std::vector<uint8_t> poly_form;// per slot: `cols` holds the COEFFICIENTS (streamed cosets): the sweep builds coset b of it per unit (Backend::coset_of)
    std::vector<Fe> chal;
};
// how many constraint terms go into one sweep kernel (EZKL_PROVER_SWEEP_TERMS; 0 = all in one kernel)
static uint32_t sweep_terms_per_program() {
    static const uint32_t v = [] {
        const char* e = getenv("EZKL_PROVER_SWEEP_TERMS");
        const int x = e ? atoi(e) : -1;
        return (uint32_t)(x >= 0 ? x : 0);
    }();
    return v;
}
// ... and how many INSTRUCTIONS at most (EZKL_PROVER_SWEEP_INSTRS, default 640; 0 = no limit): hiprtc's time grows much faster than the
// program -- the 529-instruction sweep of the k = 20 MLP compiles in 1.6 s, the ~1300-instruction one of the k = 22 / 30-column circuit took
// 50 s of a 51 s key generation (profiles/r03y_mlp_k22_30cols_full.log), and hiprtc serialises concurrent compilations (4 threads: 4.7,
// 9.7, 14.2, 18.5 s), so threads do not help -- while a sweep cut at term boundaries only re-reads the columns two kernels share.  A
// program is closed once it has reached the limit (it may exceed it by its last emitter).
static uint32_t sweep_instrs_per_program() {
    static const uint32_t v = [] {
        const char* e = getenv("EZKL_PROVER_SWEEP_INSTRS");
        const int x = e ? atoi(e) : -1;
        return (uint32_t)(x >= 0 ? x : 640);
    }();
    return v;
}
// Straight-line programs over the extended-coset columns: custom gates, then the permutation and lookup constraints, folded with y
// (value = value*y + constraint), as Evaluator::evaluate_h does.  The programs are written for ONE coset (k = ext_k = cs.k: a rotation
// by r is a shift by r rows inside the coset) and run once per coset of the extended domain.  The Horner fold starts from
// ValueSource::PreviousValue, so the list of terms can be cut anywhere: kernel p+1 continues where kernel p stopped (smaller kernels
// compile much faster -- hiprtc time grows faster than linearly with the program -- and each reads only the columns of its own terms).
// *_owner: owner rank of the witness columns (advice by column, z by chunk, m / phi by lookup), -1 = every rank holds them.
static Quotient quotient_program(const ConstraintSystem& cs, const ProvingKey& pk, const std::vector<Col>& adv_cosets, const std::vector<Col>& z_cosets,
                                 const Fe& beta, const Fe& gamma, const Fe& y, const Fe& theta, const std::vector<Col>& m_cosets,
                                 const std::vector<Col>& phi_cosets, const std::vector<Col>& inst_cosets, const std::vector<Fe>& user_chal,
                                 const std::vector<int>& adv_owner = {}, const std::vector<int>& z_owner = {}, const std::vector<int>& lk_owner = {},
                                 int inst_owner = -1) {
    // pk.stream: the caller hands COEFFICIENT forms in adv_cosets / z_cosets / m_cosets / phi_cosets, the key's come from *_polys
    const bool stream = pk.stream;
    Quotient Q{{}, {}, {}, {}, {}, {y, beta, gamma}};
    Q.chal.insert(Q.chal.end(), user_chal.begin(), user_chal.end());
    std::map<std::vector<uint32_t>, uint32_t> index;
    auto own = [](const std::vector<int>& v, uint32_t i) { return i < v.size() ? v[i] : -1; };
    auto slot = [&](std::vector<uint32_t> name, const Col& h, int owner = -1, bool is_key = false, bool poly_form = false) {
        auto it = index.find(name);
        if (it != index.end()) return it->second;
        index[name] = (uint32_t)Q.cols.size();
        Q.cols.push_back(h);
        Q.owner.push_back(owner);
        Q.key.push_back(is_key ? 1 : 0);
        Q.poly_form.push_back(poly_form ? 1 : 0);
        return (uint32_t)Q.cols.size() - 1;
    };
    enum : uint32_t { S_L0 = 100, S_LLAST, S_LACT, S_X, S_Z, S_SIGMA, S_PHI, S_M };
    auto col_slot = [&](uint32_t kind, uint32_t c) {
        return kind == N_ADV    ? slot({kind, c}, adv_cosets[c], own(adv_owner, c), false, stream)
               : kind == N_INST ? slot({kind, c}, inst_cosets[c], inst_owner)
                                : slot({kind, c}, stream ? pk.fixed_polys[c] : pk.fixed_cosets[c], -1, true, stream);
    };
    // an emitter appends the terms of one gate / one permutation chunk / one lookup argument to the program it is handed; sub-expressions
    // are shared inside an emitter's program (Lowering's memo), never across programs
    using Emit = std::function<void(Program&, Lowering&, std::vector<Src>&)>;
    std::vector<std::pair<Emit, uint32_t>> emitters;          // (emitter, number of terms it appends)
    for (uint32_t g : cs.gates) emitters.push_back({[g](Program&, Lowering& low, std::vector<Src>& terms) { terms.push_back(low.lower(g)); }, 1});
    if (!cs.perm.empty()) {
        const uint32_t s_l0 = slot({S_L0}, pk.l0, -1, true), s_ll = slot({S_LLAST}, pk.l_last, -1, true), s_la = slot({S_LACT}, pk.l_active, -1, true),
                       s_x = slot({S_X}, pk.x_coset, -1, true);
        const uint32_t nz = (uint32_t)z_cosets.size();
        std::vector<uint32_t> zc;
        for (uint32_t j = 0; j < nz; j++) zc.push_back(slot({S_Z, j}, z_cosets[j], own(z_owner, j), false, stream));
        const uint32_t usable = cs.usable;
        emitters.push_back({[=](Program& prog, Lowering&, std::vector<Src>& terms) {
                                const Src l0 = prog.column(s_l0), llast = prog.column(s_ll), one = prog.constant(Fe::one());
                                terms.push_back(prog.mul(l0, prog.sub(one, prog.column(zc[0]))));
                                const Src zl = prog.column(zc[nz - 1]);
                                terms.push_back(prog.mul(llast, prog.sub(prog.calc(EZKL_OP_SQUARE, zl), zl)));
                                for (uint32_t j = 1; j < nz; j++) terms.push_back(prog.mul(l0, prog.sub(prog.column(zc[j]), prog.column(zc[j - 1], (int32_t)usable))));
                            },
                            1 + nz});
        uint32_t pos = 0, j = 0;
        const Fe delta{FR_DELTA};
        for (auto& chunk : cs.perm_chunks()) {
            std::vector<uint32_t> v_slots, s_slots, bd_idx;
            for (uint32_t i = 0; i < chunk.size(); i++) {
                v_slots.push_back(col_slot(chunk[i].first, chunk[i].second));
                s_slots.push_back(slot({S_SIGMA, pos + i}, stream ? pk.sigma_polys[pos + i] : pk.sigma_cosets[pos + i], -1, true, stream));
                Q.chal.push_back(beta * delta.pow(pos + i));
                bd_idx.push_back((uint32_t)Q.chal.size() - 1);
            }
            const uint32_t zj = zc[j];
            emitters.push_back({[=](Program& prog, Lowering&, std::vector<Src>& terms) {
                                    const Src BETA = prog.challenge(1), GAMMA = prog.challenge(2), X = prog.column(s_x), lact = prog.column(s_la);
                                    Src left = prog.column(zj, 1), right = prog.column(zj);
                                    for (size_t i = 0; i < v_slots.size(); i++) {
                                        const Src vv = prog.column(v_slots[i]), sg = prog.column(s_slots[i]);
                                        left = prog.mul(left, prog.add(prog.add(vv, prog.mul(BETA, sg)), GAMMA));
                                        right = prog.mul(right, prog.add(prog.add(vv, prog.mul(prog.challenge(bd_idx[i]), X)), GAMMA));
                                    }
                                    terms.push_back(prog.mul(lact, prog.sub(left, right)));
                                },
                                1});
            pos += (uint32_t)chunk.size();
            j++;
        }
    }
    if (!cs.lookups.empty()) {
        const uint32_t s_l0 = slot({S_L0}, pk.l0, -1, true), s_ll = slot({S_LLAST}, pk.l_last, -1, true), s_la = slot({S_LACT}, pk.l_active, -1, true);
        Q.chal.push_back(theta);
        const uint32_t theta_idx = (uint32_t)Q.chal.size() - 1;
        for (uint32_t i = 0; i < cs.lookups.size(); i++) {
            const Lookup* lk = &cs.lookups[i];
            const uint32_t phi_s = slot({S_PHI, i}, phi_cosets[i], own(lk_owner, i), false, stream),
                           m_s = slot({S_M, i}, m_cosets[i], own(lk_owner, i), false, stream);
            emitters.push_back({[=](Program& prog, Lowering& low, std::vector<Src>& terms) {
                                    const Src l0 = prog.column(s_l0), llast = prog.column(s_ll), lact = prog.column(s_la), BETA = prog.challenge(1),
                                              THETA = prog.challenge(theta_idx);
                                    const Src phi = prog.column(phi_s), phi_next = prog.column(phi_s, 1), mcol = prog.column(m_s);
                                    std::vector<Src> fb;
                                    for (auto& t : lk->inputs) fb.push_back(prog.add(low.compress(t, THETA), BETA));
                                    const Src tb = prog.add(low.compress(lk->table, THETA), BETA);
                                    Src prodf = fb[0];
                                    for (size_t f = 1; f < fb.size(); f++) prodf = prog.mul(prodf, fb[f]);
                                    Src ssum{};                                       // sum_j prod_{i != j} (f_i + beta)
                                    for (size_t jj = 0; jj < fb.size(); jj++) {
                                        bool have = false;
                                        Src pj{};
                                        for (size_t i2 = 0; i2 < fb.size(); i2++) {
                                            if (i2 == jj) continue;
                                            pj = have ? prog.mul(pj, fb[i2]) : fb[i2];
                                            have = true;
                                        }
                                        if (!have) pj = prog.constant(Fe::one());
                                        ssum = jj == 0 ? pj : prog.add(ssum, pj);
                                    }
                                    const Src lhs = prog.mul(prog.mul(prog.sub(phi_next, phi), prodf), tb);
                                    const Src rhs = prog.sub(prog.mul(ssum, tb), prog.mul(mcol, prodf));
                                    terms.push_back(prog.mul(l0, phi));
                                    terms.push_back(prog.mul(llast, phi));
                                    terms.push_back(prog.mul(lact, prog.sub(lhs, rhs)));
                                },
                                3});
        }
    }
    // cut the emitter list into programs of at most `limit` terms (0: one program)
    const uint32_t limit = sweep_terms_per_program();
    size_t e0 = 0;
    while (e0 < emitters.size() || Q.progs.empty()) {
        Q.progs.emplace_back(cs.k, cs.k);
        Program& prog = Q.progs.back();
        Lowering low{cs, prog, col_slot, [&prog](uint32_t idx) { return prog.challenge(3 + idx); }, {}};
        std::vector<Src> terms;
        uint32_t count = 0;
        const uint32_t instr_limit = sweep_instrs_per_program();
        while (e0 < emitters.size() && (count == 0 || limit == 0 || count + emitters[e0].second <= limit)) {
            emitters[e0].first(prog, low, terms);
            count += emitters[e0].second;
            e0++;
            if (instr_limit && prog.code.size() / 8 + terms.size() >= instr_limit) break;
        }
        prog.horner(prog.previous(), terms, prog.challenge(0));
        if (emitters.empty()) break;
    }
    return Q;
}
// keygen / key loading: have the sweep kernels of this circuit compiled (and on disk) before the first proof asks for them.  The programs
// are a function of the constraint system alone (challenges and columns are run-time operands), so what is built here with placeholder
// columns has the code bytes create_proof will build.  Best effort: a failure only means the first proof compiles it itself.
static void prepare_quotient(const ProvingKey& pk) {
    const ConstraintSystem& cs = *pk.cs;
    try {
        std::vector<Col> adv(cs.n_advice), zc(cs.n_chunks), mc(cs.lookups.size()), pc(cs.lookups.size()), ic(cs.n_instance);
        std::vector<Fe> uc(cs.n_challenges, Fe::zero());
        Quotient Q = quotient_program(cs, pk, adv, zc, Fe::one(), Fe::one(), Fe::one(), Fe::one(), mc, pc, ic, uc);
        for (auto& prog : Q.progs) prog.prepare(Q.cols.size(), Q.chal.size());
    } catch (const Error& e) {
        if (getenv("EZKL_HIP_JIT_DEBUG")) fprintf(stderr, "[ezkl_prover] sweep kernel not prepared at keygen: %s\n", e.what());
    }
}
// what one launch of this key's sweep does per row: [instructions, Montgomery products (MUL, SQUARE, HORNER_STEP), column slots, programs]
static void sweep_stats(const ProvingKey& pk, uint64_t out[4]) {
    const ConstraintSystem& cs = *pk.cs;
    std::vector<Col> adv(cs.n_advice), zc(cs.n_chunks), mc(cs.lookups.size()), pc(cs.lookups.size()), ic(cs.n_instance);
    std::vector<Fe> uc(cs.n_challenges, Fe::zero());
    Quotient Q = quotient_program(cs, pk, adv, zc, Fe::one(), Fe::one(), Fe::one(), Fe::one(), mc, pc, ic, uc);
    out[0] = out[1] = 0;
    for (auto& prog : Q.progs)
        for (size_t i = 0; i + 8 <= prog.code.size(); i += 8) {
            out[0]++;
            const uint32_t op = prog.code[i];
            if (op == EZKL_OP_MUL || op == EZKL_OP_SQUARE || op == EZKL_OP_HORNER_STEP) out[1]++;
        }
    out[2] = Q.cols.size();
    out[3] = Q.progs.size();
}
// the theta-compressed lookup column over the n rows of the Lagrange domain (a gate program with ext_k = k)
static Col compress_column(const ConstraintSystem& cs, const Backend& be, const std::vector<uint32_t>& tuple, const Fe& theta,
                           const std::function<Col(uint32_t, uint32_t)>& col_handle, const std::vector<Fe>& user_chal) {
    Program prog(cs.k, cs.k);
    std::vector<Col> cols;
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> index;
    Lowering low{cs, prog,
                 [&](uint32_t kind, uint32_t c) {
                     auto key = std::make_pair(kind, c);
                     auto it = index.find(key);
                     if (it != index.end()) return it->second;
                     index[key] = (uint32_t)cols.size();
                     cols.push_back(col_handle(kind, c));
                     return (uint32_t)cols.size() - 1;
                 },
                 [&](uint32_t idx) { return prog.challenge(1 + idx); },
                 {}};
    Src r = low.compress(tuple, prog.challenge(0));
    if (r.kind == EZKL_SRC_COLUMN && prog.rotations[r.rot] == 0) return cols[r.idx];   // a bare column (a lookup table): it IS its compression, read-only below
    if (r.kind != EZKL_SRC_INTERMEDIATE) prog.calc(EZKL_OP_STORE, r);        // a rotated column / constant: materialise it
    Col out = be.zeros(cs.n);
    std::vector<Fe> chal = {theta};
    chal.insert(chal.end(), user_chal.begin(), user_chal.end());
    prog.run(cols, chal, out->ptr());
    return out;
}

// ------------------------------------------------------------------ create_proof
struct Stopwatch {
    double* out;
    std::chrono::steady_clock::time_point t0, start;
    explicit Stopwatch(double* o) : out(o), t0(std::chrono::steady_clock::now()), start(t0) {
        if (out) std::fill(out, out + 12, 0.0);
    }
    void lap(int i) {
        auto now = std::chrono::steady_clock::now();
        if (out) out[i] += std::chrono::duration<double>(now - t0).count();
        t0 = now;
    }
    void total() {
        if (out) out[10] = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    }
};
// the centred representative of a rotation inside a coset of n rows: in (-n/2, n/2]
static int64_t centred(int64_t rot, int64_t n) {
    int64_t r = ((rot % n) + n) % n;
    return r > n / 2 ? r - n : r;
}
static std::vector<uint8_t> create_proof(ProvingKey& pk, ezkl_bases_t g, ezkl_bases_t gl, const void* const* advice, ezkl_advice_fn advice_fn, void* advice_user,
                                         const void* const* instances, const uint32_t* instance_lens, Rng& rng, double* timings,
                                         const uint8_t* advice_formats = nullptr) {
    ConstraintSystem& cs = *pk.cs;
    const uint32_t n = cs.n, k = cs.k, u = cs.usable;
    // the helper chains of a proof are hundreds of small device-only calls: queued on the library stream without a host round trip
    // each (ezkl_hip_set_async); every call that returns host data still synchronises by itself.  EZKL_PROVER_SYNC_CALLS=1: off.
    struct AsyncCalls {
        int prev = 0;
        bool on = false;
        AsyncCalls() {
            if (getenv("EZKL_PROVER_SYNC_CALLS")) return;
            on = ezkl_hip_set_async(1, &prev) == EZKL_OK;
        }
        ~AsyncCalls() {
            if (on) (void)ezkl_hip_set_async(prev, nullptr);
        }
    } async_calls;
    Backend be(k, n, g, gl, cs.shard);
    be.recv_bytes = &cs.shard.stats[2];
    const Topo& topo = be.topo;
    const bool owners = topo.owners;
    invalid(pk.stream && (owners || cs.shard.on()), "a streamed key (EZKL_KEY_COSETS) proves on one rank");
    be.stream_cosets = pk.stream;
    for (auto& x : cs.shard.stats) x = 0;
    Stopwatch sw(timings);
    EvmTranscript T;
    T.common_scalar(pk.digest);
    // 0. instances: absorbed, never committed (halo2 KZG: QUERY_INSTANCE = false)
    std::vector<Col> inst_cols;
    for (uint32_t i = 0; i < cs.n_instance; i++) {
        invalid(instance_lens[i] > u, "too many instance values");
        std::vector<U256> vals(instance_lens[i]);
        for (uint32_t j = 0; j < instance_lens[i]; j++) {
            std::memcpy(vals[j].data(), (const uint8_t*)instances[i] + 32 * j, 32);
            invalid(cmp(vals[j], FR.p) >= 0, "non-canonical instance value");
            T.common_scalar(Fe{vals[j]});
        }
        // the column is zero outside its few public values: filled on the device, only the values cross PCIe (a zeroed 2^k-row host
        // vector + a pageable 32 MiB copy cost ~5 ms of host time at k = 20 before the first advice column could move)
        Col col = be.zeros(n);
        be.set_rows(col, 0, vals);
        inst_cols.push_back(col);
    }
    // 5'. The vanishing argument's random polynomial depends on nothing but the randomness, and the library's generator addresses its
    //     requests by index (request number k = ChaCha stream k): its column is expanded and its commitment STARTED here, so that the one
    //     uniform-scalar MSM of that step (≈ 1.5 ms + its stage's round trips) runs under the upload of the witness, when the GPU has
    //     little else to do; step 5 below collects the point and writes it at the same place of the transcript, drawn at the same place of
    //     the random stream.  One prover on one GPU with the library's generator only (a caller's rng callback is sequential; a sharded
    //     prover commits this column by point ranges).  EZKL_PROVER_NO_EARLY_RANDOM=1: off.
    struct EarlyRandom {
        Col col;
        int token = -1;
        uint64_t stream = 0;
        ~EarlyRandom() {                             // unwinding: give the call slot back
            if (token >= 0) { G1 sink; (void)ezkl_hip_msm_g1_finish(token, &sink); }
        }
    } early;
    if (!rng.fn && !cs.shard.on() && !getenv("EZKL_PROVER_NO_EARLY_RANDOM")) {
        uint64_t draws = 0;                          // the requests that precede it: blinding rows of advice / m / z / phi
        for (uint32_t c = 0; c < cs.n_advice; c++) draws += cs.unblinded[c] ? 0 : 1;
        draws += 2 * (uint64_t)cs.lookups.size() + cs.n_chunks;
        early.stream = rng.calls + draws;
        early.col = be.alloc(n);
        check(ezkl_hip_chacha20_fr_dev(rng.key, early.stream, 0, early.col->ptr(), n, nullptr), "ezkl_hip_chacha20_fr_dev");
        check(ezkl_hip_msm_g1_start_dev(g, 0, early.col->ptr(), n, &early.token), "ezkl_hip_msm_g1_start_dev");
    }
    // 1. advice columns, phase by phase; the phase-0 commitments seed the user challenges.  Every rank uploads every column (the
    //    lookup / permutation arguments it owns read them); in owner mode only the owner transforms and commits a column.
    std::vector<Col> adv_cols(cs.n_advice);
    std::vector<Backend::Forms> adv_forms(cs.n_advice);
    std::vector<int> adv_owner(cs.n_advice, -1);
    std::vector<Fe> user_chal;
    // owner mode: which advice columns THIS rank reads in Lagrange form at all -- the ones it owns (forms + commitment), the ones the
    // lookup arguments it owns compress, the ones in the permutation chunks it owns.  Only those cross its PCIe link (at 8 ranks about
    // half of the k = 20 MLP's witness: the upload is the one stage every rank would otherwise repeat in full).
    std::vector<uint8_t> adv_needed(cs.n_advice, owners ? 0 : 1);
    if (owners) {
        std::vector<uint8_t> seen(cs.nodes.size(), 0);
        std::set<Query> qs_[3];
        for (size_t i = 0; i < cs.lookups.size(); i++) {
            if (!topo.mine(i)) continue;
            for (auto& t : cs.lookups[i].inputs)
                for (uint32_t e : t) cs.collect(e, qs_, seen);
            for (uint32_t e : cs.lookups[i].table) cs.collect(e, qs_, seen);
        }
        for (auto& q : qs_[0]) adv_needed[q.col] = 1;
        size_t j = 0;
        for (auto& chunk : cs.perm_chunks()) {
            if (topo.mine(j))
                for (auto& pc : chunk)
                    if (pc.first == N_ADV) adv_needed[pc.second] = 1;
            j++;
        }
    }
    for (uint32_t phase = 0; phase < 2; phase++) {
        std::vector<uint32_t> idxs;
        for (uint32_t c = 0; c < cs.n_advice; c++)
            if (cs.advice_phase[c] == phase) idxs.push_back(c);
        if (idxs.empty()) continue;
        std::vector<std::vector<U256>> host;
        std::vector<const void*> src(cs.n_advice, nullptr);
        if (advice_fn) {
            std::vector<void*> dst(cs.n_advice, nullptr);
            if (!cs.advice_by_pointer) {
                host.resize(cs.n_advice);
                for (uint32_t c : idxs) {
                    host[c].assign(n, U256{0, 0, 0, 0});
                    dst[c] = host[c].data();
                }
            }
            std::vector<U256> ch;
            for (auto& f : user_chal) ch.push_back(f.v);
            invalid(advice_fn(advice_user, phase, ch.data(), (uint32_t)ch.size(), dst.data()) != 0, "advice callback failed");
            for (uint32_t c : idxs) {
                invalid(dst[c] == nullptr, "advice callback left a column of this phase unset");
                src[c] = dst[c];                              // by pointer: the callee's own buffers, valid until create_proof returns
            }
        } else {
            invalid(advice == nullptr, "no advice columns");
            for (uint32_t c : idxs) src[c] = advice[c];
        }
        // upload, blind and commit the phase in ONE call: all copies go to a copy stream at once and the MSM of a column waits
        // only for its own copy, so PCIe runs under the kernels (a blocking upload per column followed by the batch costs the sum)
        std::vector<std::vector<U256>> tails;
        std::vector<const void*> hostp, tailp;
        std::vector<void*> devp;
        std::vector<size_t> up_of(idxs.size(), SIZE_MAX);       // position of column j of the phase in the upload (owner mode: only what this rank reads)
        for (size_t j = 0; j < idxs.size(); j++) {
            const uint32_t c = idxs[j];
            invalid(src[c] == nullptr, "missing advice column");
            if (owners) adv_owner[c] = (int)topo.owner(j);
            std::vector<U256> tail = cs.unblinded[c] ? std::vector<U256>(n - u, Fe::one().v)     // Blind::default() (polycommit.rs:57-61), no randomness drawn
                                                     : rng.vec(n - u);                           // blinding rows [u, n): drawn on every rank, same stream
            if (!(topo.mine(j) || adv_needed[c])) continue;
            adv_cols[c] = be.alloc(n);
            up_of[j] = hostp.size();
            tails.push_back(std::move(tail));
            hostp.push_back(src[c]);
            devp.push_back(adv_cols[c]->ptr());
        }
        for (auto& t : tails) tailp.push_back(t.data());
        // what each host column holds (ezkl_hip_upload_begin_fmt): 32-byte Fp words, or the IntegerRep values they were made from
        std::vector<uint8_t> fmts;
        if (advice_formats) {
            for (size_t j = 0; j < idxs.size(); j++) {
                if (up_of[j] == SIZE_MAX) continue;
                const uint8_t f = advice_formats[idxs[j]];
                invalid(f > EZKL_COLUMN_INT128, "unknown advice column format");
                invalid(f != EZKL_COLUMN_FP && advice_fn && !cs.advice_by_pointer, "integer advice columns need caller-owned buffers (direct pointers, or a by-pointer callback)");
                fmts.push_back(f);
            }
        }
        std::vector<G1> commits(idxs.size());
        const bool by_batch = cs.shard.on() && cs.shard.full_bases;       // sharded with complete base sets: commit after the copies have landed
        {
            // the phase in steps (ezkl_hip_upload_commit_batch in one call): every copy is queued, the NTTs of column j are queued
            // behind ITS copy on the aux stream, then the commits run -- PCIe, MSMs and NTTs overlap
            ezkl_upload_t up = nullptr;
            check(ezkl_hip_upload_begin_fmt(hostp.data(), fmts.empty() ? nullptr : fmts.data(), devp.data(), hostp.size(), n, tailp.data(), u, n - u, &up),
                  "ezkl_hip_upload_begin_fmt");
            int rc = EZKL_OK;
            for (size_t j = 0; j < idxs.size(); j++)
                if (topo.mine(j)) adv_forms[idxs[j]] = be.forms_alloc(cs.ext_k);     // before the copies are in flight
            try {
                for (size_t j = 0; j < idxs.size(); j++) {
                    if (!topo.mine(j)) continue;
                    check(ezkl_hip_upload_wait(up, up_of[j], be.aux_stream()), "ezkl_hip_upload_wait");
                    adv_forms[idxs[j]] = be.forms_async(adv_cols[idxs[j]], cs.ext_k, &adv_forms[idxs[j]]);
                    cs.shard.stats[0]++;
                }
                if (!by_batch) rc = ezkl_hip_upload_commit(up, gl, be.commit_first(), be.commit_count(), commits.data());
            } catch (...) {
                (void)ezkl_hip_upload_end(up);
                throw;
            }
            const int rc2 = ezkl_hip_upload_end(up);
            check(rc, "ezkl_hip_upload_commit");
            check(rc2, "ezkl_hip_upload_end");
        }
        if (by_batch) {                                     // divided by columns / by owner: the copies have landed (upload_end drains them)
            std::vector<Col> cols_;
            for (size_t j = 0; j < idxs.size(); j++) cols_.push_back(topo.mine(j) ? adv_cols[idxs[j]] : Col());
            commits = be.commit_columns(gl, cols_, true);
        } else
            be.fold(commits);
        for (auto& p : commits) T.write_point(p);
        if (phase == 0)
            for (uint32_t i = 0; i < cs.n_challenges; i++) user_chal.push_back(T.squeeze_challenge());
    }
    cs.shard.stats[1] += cs.n_advice;
    sw.lap(0);
    auto col_handle = [&](uint32_t kind, uint32_t c) -> Col { return kind == N_ADV ? adv_cols[c] : kind == N_INST ? inst_cols[c] : pk.fixed_values[c]; };
    // 2. theta; mv-lookup multiplicities m(X): argument i on its owner
    Fe theta = Fe::zero();
    struct LookupState {
        std::vector<Col> inputs;
        Col table, m, phi;
        Backend::Forms m_forms, phi_forms;
        bool mine = true;
    };
    const size_t nl = cs.lookups.size();
    std::vector<LookupState> lk(nl);
    std::vector<int> lk_owner(nl, -1);
    if (nl) {
        theta = T.squeeze_challenge();
        const Col missing = be.zeros(1);
        std::vector<std::vector<U256>> blinds(nl);
        std::vector<std::vector<Col>> my_inputs;
        std::vector<Col> my_tables;
        std::vector<size_t> my_idx;
        for (size_t i = 0; i < nl; i++) {
            const Lookup& l = cs.lookups[i];
            LookupState& st = lk[i];
            st.mine = topo.mine(i);
            if (owners) lk_owner[i] = (int)topo.owner(i);
            blinds[i] = rng.vec(n - u);                                 // every rank draws every argument's randomness: one stream, same order
            if (!st.mine) continue;
            for (auto& t : l.inputs) st.inputs.push_back(compress_column(cs, be, t, theta, col_handle, user_chal));
            st.table = compress_column(cs, be, l.table, theta, col_handle, user_chal);
            my_inputs.push_back(st.inputs); my_tables.push_back(st.table); my_idx.push_back(i);
            cs.shard.stats[3]++;
        }
        // the arguments are independent: their hash-table passes run as ONE batch (three launches, not three per argument)
        const std::vector<Col> my_ms = be.lookup_multiplicities(my_inputs, my_tables, u, missing);
        for (size_t q = 0; q < my_idx.size(); q++) {
            lk[my_idx[q]].m = my_ms[q];
            be.set_rows(my_ms[q], u, blinds[my_idx[q]]);
        }
        std::vector<Col> ms;
        for (auto& st : lk) ms.push_back(st.m);
        for (auto& st : lk)
            if (st.mine) { st.m_forms = be.forms_async(st.m, cs.ext_k); cs.shard.stats[0]++; }
        const auto m_commits = be.commit_columns(gl, ms, true);
        be.lookup_check(missing);                                       // after the commit's own synchronisation: no extra round trip
        for (auto& p : m_commits) T.write_point(p);
        cs.shard.stats[1] += 2 * nl;
    }
    sw.lap(1);
    // 3. beta, gamma
    const Fe beta = T.squeeze_challenge(), gamma = T.squeeze_challenge();
    // 4. permutation grand products, chained across chunks: chunk j on its owner
    // (The z could be committed WHILE the lookup arguments' helper chains run -- no challenge is drawn between the two commit batches.  Built and
    // measured in round 4: 84.3-86.5 ms merged against 83.1-84.8 ms one phase after the other on the k = 20 MLP, same bytes: the MSMs and the
    // helper chains compete for the same integer ALUs.  The experiment is gone from the source; NOTEBOOK.md §4.1.2 has it.)
    std::vector<Col> zs;
    std::vector<Backend::Forms> z_forms;
    std::vector<int> z_owner;
    {
        uint32_t pos = 0;
        std::vector<std::vector<Col>> vals_all, sigs_all;
        for (auto& chunk : cs.perm_chunks()) {
            std::vector<Col> vals, sigs;
            for (auto& pc : chunk) vals.push_back(col_handle(pc.first, pc.second));
            for (size_t i = 0; i < chunk.size(); i++) sigs.push_back(pk.sigma_values[pos + i]);
            vals_all.push_back(vals);
            sigs_all.push_back(sigs);
            pos += (uint32_t)chunk.size();
        }
        zs = be.permutation_products(vals_all, sigs_all, beta, gamma, pk.omega_col, u);
        z_forms.resize(zs.size());
        for (size_t j = 0; j < zs.size(); j++) {
            z_owner.push_back(owners ? (int)topo.owner(j) : -1);
            const std::vector<U256> blind = rng.vec(n - u - 1);
            if (zs[j]) be.set_rows(zs[j], u + 1, blind);
        }
        for (size_t j = 0; j < zs.size(); j++)
            if (zs[j]) { z_forms[j] = be.forms_async(zs[j], cs.ext_k); cs.shard.stats[0]++; cs.shard.stats[3]++; }
        for (auto& p : be.commit_columns(gl, zs, false)) T.write_point(p);
        cs.shard.stats[1] += zs.size();
    }
    sw.lap(2);
    // 4b. mv-lookup running sums phi(X): on the owner of the argument
    {
        std::vector<Col> phis(nl);
        {
            std::vector<std::vector<Col>> ins;
            std::vector<Col> tabs, ms_;
            std::vector<size_t> which;
            for (size_t i = 0; i < nl; i++)
                if (lk[i].mine) { ins.push_back(lk[i].inputs); tabs.push_back(lk[i].table); ms_.push_back(lk[i].m); which.push_back(i); }
            std::vector<Col> sums = be.lookup_grand_sums(ins, tabs, ms_, beta);
            for (size_t q = 0; q < which.size(); q++) lk[which[q]].phi = sums[q];
            for (size_t i = 0; i < nl; i++) {
                const std::vector<U256> blind = rng.vec(n - u - 1);       // same draw order as one lookup after the other
                if (lk[i].mine) be.set_rows(lk[i].phi, u + 1, blind);
                phis[i] = lk[i].phi;
            }
        }
        for (auto& st : lk)
            if (st.mine) { st.phi_forms = be.forms_async(st.phi, cs.ext_k); cs.shard.stats[0]++; }
        for (auto& p : be.commit_columns(gl, phis, false)) T.write_point(p);
    }
    sw.lap(3);
    // 5. vanishing argument: random polynomial (every rank expands the same keystream: a replicated column);  6. y
    Col rnd;
    if (early.token >= 0) {                          // started under the witness upload (step 5' above)
        invalid(rng.calls != early.stream, "internal: the random polynomial was expanded at another place of the random stream");
        rng.calls++;
        rnd = early.col;
        G1 pt;
        const int tok = early.token;
        early.token = -1;
        check(ezkl_hip_msm_g1_finish(tok, &pt), "ezkl_hip_msm_g1_finish");
        T.write_point(pt);
    } else {
        rnd = rng.column(be, n);
        T.write_point(be.commit({rnd})[0]);
    }
    const Fe y = T.squeeze_challenge();
    sw.lap(4);
    // 7. quotient
    const uint32_t log_e = cs.ext_k - k, E = 1u << log_e;
    const size_t ne = (size_t)1 << cs.ext_k;
    const int inst_owner = owners ? 0 : -1;
    std::vector<Col> adv_polys(cs.n_advice), adv_cosets(cs.n_advice), inst_cosets, z_polys(zs.size()), z_cosets(zs.size()), m_polys(nl), phi_polys(nl), m_cosets(nl),
        phi_cosets(nl);
    for (auto& h : inst_cols) inst_cosets.push_back((!owners || topo.rank == 0) ? be.coeff_to_extended(be.lagrange_to_coeff(h), cs.ext_k) : Col());
    be.aux_sync();                                   // the forms queued behind each finished column (Backend::forms_async)
    for (uint32_t c = 0; c < cs.n_advice; c++) { adv_polys[c] = adv_forms[c].poly; adv_cosets[c] = adv_forms[c].coset; }
    for (size_t j = 0; j < zs.size(); j++) { z_polys[j] = z_forms[j].poly; z_cosets[j] = z_forms[j].coset; }
    for (size_t i = 0; i < nl; i++) { m_polys[i] = lk[i].m_forms.poly; m_cosets[i] = lk[i].m_forms.coset; phi_polys[i] = lk[i].phi_forms.poly; phi_cosets[i] = lk[i].phi_forms.coset; }
    sw.lap(5);
    Col hnum = be.zeros(ne);
    {
        Quotient Q = pk.stream ? quotient_program(cs, pk, adv_polys, z_polys, beta, gamma, y, theta, m_polys, phi_polys, inst_cosets, user_chal, adv_owner, z_owner,
                                                  lk_owner, inst_owner)
                               : quotient_program(cs, pk, adv_cosets, z_cosets, beta, gamma, y, theta, m_cosets, phi_cosets, inst_cosets, user_chal, adv_owner, z_owner,
                                                  lk_owner, inst_owner);
        // The sweep runs in UNITS of rows of one coset.  One rank: the E cosets.  Sharded by rows (set_sweep_gather, equal power-of-two
        // slices): max(E, world) units -- rank r sweeps E / world whole cosets, or, with more ranks than cosets, one of the
        // world / E row ranges of a coset -- and h is all_gathered (units are in rank order, so the shards are equal slices of the
        // coset-major h).  Columns resident on this rank are read in place (a row range whose rotated window wraps around the coset
        // is stitched); in owner mode the columns of other ranks arrive through ONE exchange, one slab of rows (+ the halo its
        // rotations reach) per (unit, column).
        const bool shard_sweep = cs.shard.on() && cs.shard.gather && topo.world > 1 && (topo.world <= E || (topo.world / E) <= n / 2);
        invalid(owners && !shard_sweep, "owner mode needs the row-sharded sweep");
        const uint32_t split = shard_sweep && topo.world > E ? topo.world / E : 1;        // row ranges per coset
        uint32_t log_split = 0;
        while ((1u << log_split) < split) log_split++;
        const uint32_t len = n / split, n_units = E * split;
        auto unit_rank = [&](uint32_t un) { return shard_sweep ? un * topo.world / n_units : topo.rank; };
        const size_t ns = Q.cols.size();
        // halo of every slot: the rotations the program reads it with, centred
        std::vector<int64_t> hn(ns, 0), hp(ns, 0);
        for (auto& prog : Q.progs)
            for (size_t i = 0; i < prog.code.size(); i += 8)
                for (size_t sidx : {(size_t)2, (size_t)5})
                    if (prog.code[i + sidx] == EZKL_SRC_COLUMN) {
                        const uint32_t sl = prog.code[i + sidx + 1];
                        const int64_t r = centred(prog.rotations[prog.code[i + sidx + 2]], n);
                        hn[sl] = std::max(hn[sl], -r);
                        hp[sl] = std::max(hp[sl], r);
                    }
        auto remote = [&](size_t sl) { return owners && Q.owner[sl] >= 0 && (uint32_t)Q.owner[sl] != topo.rank; };
        // first row of coset b inside a resident column: key columns hold cosets [pk.coset_first, +pk.coset_count) only
        auto coset_row = [&](size_t sl, uint32_t b) -> size_t {
            if (!Q.key[sl]) return (size_t)b * n;
            invalid(b < pk.coset_first || b >= pk.coset_first + pk.coset_count, "the proving key was loaded for another sharding (it does not hold this coset)");
            return (size_t)(b - pk.coset_first) * n;
        };
        // slabs of the columns other ranks own, one per (my unit, remote slot); the exchange lists in the SAME order on both sides:
        // by receiving unit, then by slot, then by piece
        std::vector<std::vector<Col>> slab(n_units, std::vector<Col>(ns));
        if (owners && shard_sweep) {
            std::vector<ezkl_comm_seg_t> sends, recvs;
            for (uint32_t un = 0; un < n_units; un++) {
                const uint32_t b = un / split, part = un % split, dst = unit_rank(un);
                for (size_t sl = 0; sl < ns; sl++) {
                    if (Q.owner[sl] < 0) continue;
                    const uint32_t own_r = (uint32_t)Q.owner[sl];
                    if (own_r == dst) continue;                                  // read in place by its owner
                    if (own_r != topo.rank && dst != topo.rank) continue;
                    const int64_t lo = (int64_t)part * len - (split > 1 ? hn[sl] : 0), rows = (int64_t)len + (split > 1 ? hn[sl] + hp[sl] : 0);
                    invalid(rows > (int64_t)n, "rotation halo larger than a coset");
                    const int64_t start = ((lo % (int64_t)n) + n) % n, first = std::min<int64_t>(rows, (int64_t)n - start);
                    if (dst == topo.rank) {
                        slab[un][sl] = be.alloc((size_t)rows);
                        recvs.push_back({(int)own_r, slab[un][sl]->ptr(), (size_t)first * 32});
                        if (rows > first) recvs.push_back({(int)own_r, Backend::at(slab[un][sl], (size_t)first), (size_t)(rows - first) * 32});
                        cs.shard.stats[2] += (uint64_t)rows * 32;
                    } else {
                        invalid(!Q.cols[sl], "owned column missing");
                        sends.push_back({(int)dst, Backend::at(Q.cols[sl], (size_t)b * n + (size_t)start), (size_t)first * 32});
                        if (rows > first) sends.push_back({(int)dst, Backend::at(Q.cols[sl], (size_t)b * n), (size_t)(rows - first) * 32});
                    }
                }
            }
            check(ezkl_hip_synchronize(), "ezkl_hip_synchronize");               // the columns to send are complete (library + aux streams)
            invalid(cs.shard.exchange(cs.shard.xuser, sends.data(), sends.size(), recvs.data(), recvs.size()) != 0, "exchange callback failed");
        }
        std::vector<Col> stitched;
        for (uint32_t un = 0; un < n_units; un++) {
            if (unit_rank(un) != topo.rank) continue;
            const uint32_t b = un / split, part = un % split;
            const size_t rlo = (size_t)part * len;
            void* out = Backend::at(hnum, (size_t)b * n + rlo);
            if (split == 1) {                                                     // a whole coset: the programs as they are
                std::vector<const void*> ptrs;
                std::vector<Col> unit_cosets;                                     // streamed: coset b of every column that is held as coefficients
                for (size_t sl = 0; sl < ns; sl++) {
                    if (Q.poly_form[sl] && Q.cols[sl]) {
                        unit_cosets.push_back(be.coset_of(Q.cols[sl], cs.ext_k, b));
                        ptrs.push_back(unit_cosets.back()->ptr());
                        continue;
                    }
                    ptrs.push_back(remote(sl) ? slab[un][sl]->ptr() : (Q.cols[sl] ? Backend::at(Q.cols[sl], coset_row(sl, b)) : nullptr));
                }
                for (auto& prog : Q.progs) prog.run_ptrs(ptrs, Q.chal, out);
            } else {                                                              // a row range: every (column, rotation) becomes a window at rotation 0
                for (auto& prog : Q.progs) {
                    std::vector<const void*> ptrs;
                    std::vector<std::pair<uint32_t, int64_t>> windows;
                    const Program sub = prog.row_sharded(log_split, windows);
                    for (auto& w : windows) {
                        const size_t sl = w.first;
                        const int64_t sh = centred(w.second, n);
                        if (remote(sl)) {
                            ptrs.push_back(Backend::at(slab[un][sl], (size_t)(hn[sl] + sh)));
                            continue;
                        }
                        const int64_t start = ((((int64_t)rlo + sh) % (int64_t)n) + n) % n;
                        const Col& col = Q.cols[sl];
                        const size_t row0 = coset_row(sl, b);
                        if (start + (int64_t)len <= (int64_t)n) {
                            ptrs.push_back(Backend::at(col, row0 + (size_t)start));
                        } else {                                                  // the window wraps around the coset
                            Col t = be.alloc(len);
                            const size_t first = (size_t)((int64_t)n - start);
                            be.scale_into(Backend::at(col, row0 + (size_t)start), be.one, t->ptr(), first);
                            be.scale_into(Backend::at(col, row0), be.one, Backend::at(t, first), len - first);
                            stitched.push_back(t);
                            ptrs.push_back(t->ptr());
                        }
                    }
                    sub.run_ptrs(ptrs, Q.chal, out);
                }
            }
            // divide by the vanishing polynomial: a constant on the coset
            be.scale_into(out, be.vanishing_inv(cs.ext_k, b), out, len);
        }
        if (shard_sweep) {
            const size_t per = ne / topo.world * 32;
            invalid(cs.shard.gather(cs.shard.gather_user, hnum->ptr(), ne * 32, (size_t)topo.rank * per, per) != 0, "gather callback failed");
            cs.shard.sharded_sweeps++;
        }
    }
    sw.lap(6);
    adv_cosets.clear(); z_cosets.clear(); m_cosets.clear(); phi_cosets.clear(); inst_cosets.clear();
    for (auto& f : adv_forms) f.coset.reset();
    for (auto& f : z_forms) f.coset.reset();
    for (auto& st : lk) { st.m_forms.coset.reset(); st.phi_forms.coset.reset(); }
    Col hcoef = be.extended_to_coeff(hnum, cs.ext_k);
    hnum.reset();
    const uint32_t npieces = cs.degree - 1;
    std::vector<Col> pieces;
    for (uint32_t i = 0; i < npieces; i++) pieces.push_back(be.slice_copy(hcoef, (size_t)i * n, n));
    for (auto& p : be.commit(pieces)) T.write_point(p);
    hcoef.reset();
    sw.lap(7);
    // 8. x
    const Fe x = T.squeeze_challenge();
    const Fe w = omega(k);
    std::map<int32_t, Fe> rot_memo;                 // a handful of distinct rotations, hundreds of queries
    auto rot_point = [&](int32_t r) {
        auto it = rot_memo.find(r);
        if (it == rot_memo.end()) it = rot_memo.emplace(r, x * w.pow((uint64_t)(r >= 0 ? (uint32_t)r % n : n - ((uint32_t)(-r) % n)))).first;
        return it->second;
    };
    // 9. evaluations: every (polynomial, point) of this round in ONE batched call per rank, then written in transcript order.  Owner
    //    mode: a polynomial is evaluated by the rank that holds it (replicated ones -- key columns, the random polynomial, h -- are
    //    dealt round-robin) and the scalars are all_gathered.
    const Fe xn = x.pow((uint64_t)n);
    Col hcomb = be.alloc(n);                       // h(X) = sum_i x^(n i) * piece_i(X): what the verifier reconstructs from the pieces
    {
        std::vector<Fe> cf;
        Fe p = Fe::one();
        for (uint32_t i = 0; i < npieces; i++) { cf.push_back(p); p = p * xn; }
        be.lincomb(hcomb, pieces, cf, n, false);
    }
    std::vector<Col> ev_polys;
    std::vector<Fe> ev_pts;
    std::vector<uint32_t> ev_owner;
    uint32_t deal = 0;                             // round-robin over the replicated polynomials
    std::map<const void*, uint32_t> dealt;
    auto owner_of = [&](const Col& poly, int own) -> uint32_t {
        if (!owners) return topo.rank;
        if (own >= 0) return (uint32_t)own;
        auto it = dealt.find(poly.get());
        if (it == dealt.end()) it = dealt.emplace(poly.get(), deal++ % topo.world).first;
        return it->second;
    };
    auto want = [&](const Col& poly, const Fe& pt, uint32_t own) { ev_polys.push_back(poly); ev_pts.push_back(pt); ev_owner.push_back(own); return ev_polys.size() - 1; };
    const Fe x_next = rot_point(1), x_last = rot_point((int32_t)u);
    // owners of the polynomials, fixed once (evaluation and opening must agree); replicated polynomials are keyed by their column handle
    std::vector<uint32_t> fix_own(cs.n_fixed), sig_own(pk.sigma_polys.size());
    for (uint32_t c = 0; c < cs.n_fixed; c++) fix_own[c] = owner_of(pk.fixed_polys[c], -1);
    for (size_t i = 0; i < pk.sigma_polys.size(); i++) sig_own[i] = owner_of(pk.sigma_polys[i], -1);
    const uint32_t rnd_own = owner_of(rnd, -1), h_own = owner_of(hcomb, -1);
    for (auto& q : cs.advice_queries) want(adv_polys[q.col], rot_point(q.rot), owner_of(adv_polys[q.col], adv_owner[q.col]));
    for (auto& q : cs.fixed_queries) want(pk.fixed_polys[q.col], rot_point(q.rot), fix_own[q.col]);
    want(rnd, x, rnd_own);
    for (size_t i = 0; i < pk.sigma_polys.size(); i++) want(pk.sigma_polys[i], x, sig_own[i]);
    for (size_t j = 0; j < z_polys.size(); j++) {
        const uint32_t o = owner_of(z_polys[j], z_owner[j]);
        want(z_polys[j], x, o);
        want(z_polys[j], x_next, o);
        if (j + 1 < z_polys.size()) want(z_polys[j], x_last, o);
    }
    for (size_t i = 0; i < nl; i++) {
        const uint32_t o = owner_of(phi_polys[i], lk_owner[i]);
        want(phi_polys[i], x, o);                  // mv_lookup::prover::Committed::evaluate: phi(x), phi(wx), m(x)
        want(phi_polys[i], x_next, o);
        want(m_polys[i], x, o);
    }
    const size_t n_written = ev_polys.size();
    const size_t h_slot = want(hcomb, x, h_own);          // not part of the proof: the verifier derives it
    std::vector<Fe> ev(ev_polys.size(), Fe::zero());
    {
        std::vector<Col> my_polys;
        std::vector<Fe> my_pts;
        std::vector<size_t> my_idx;
        for (size_t i = 0; i < ev_polys.size(); i++)
            if (ev_owner[i] == topo.rank) { my_polys.push_back(ev_polys[i]); my_pts.push_back(ev_pts[i]); my_idx.push_back(i); }
        const std::vector<Fe> mine = be.eval_poly_batch(my_polys, my_pts, n);
        if (!owners) {
            for (size_t q = 0; q < my_idx.size(); q++) ev[my_idx[q]] = mine[q];
        } else {
            const size_t per = ev_polys.size();
            std::vector<Fe> all((size_t)topo.world * per, Fe::zero());
            for (size_t q = 0; q < my_idx.size(); q++) all[(size_t)topo.rank * per + my_idx[q]] = mine[q];
            be.allgather_fe(all, per);
            for (size_t i = 0; i < per; i++) ev[i] = all[(size_t)ev_owner[i] * per + i];
        }
    }
    for (size_t i = 0; i < n_written; i++) T.write_scalar(ev[i]);
    size_t cursor = 0;
    std::map<std::pair<uint32_t, int32_t>, Fe> adv_evals, fix_evals;
    for (auto& q : cs.advice_queries) adv_evals[{q.col, q.rot}] = ev[cursor++];
    for (auto& q : cs.fixed_queries) fix_evals[{q.col, q.rot}] = ev[cursor++];
    const Fe random_eval = ev[cursor++];
    std::vector<Fe> sigma_evals;
    for (size_t i = 0; i < pk.sigma_polys.size(); i++) sigma_evals.push_back(ev[cursor++]);
    struct ZEval {
        Fe e0, e1, e2;
        bool has2;
    };
    std::vector<ZEval> z_evals;
    for (size_t j = 0; j < z_polys.size(); j++) {
        ZEval ze{ev[cursor], ev[cursor + 1], Fe::zero(), j + 1 < z_polys.size()};
        cursor += 2;
        if (ze.has2) ze.e2 = ev[cursor++];
        z_evals.push_back(ze);
    }
    std::vector<std::array<Fe, 3>> lk_evals;
    for (size_t i = 0; i < nl; i++) {
        lk_evals.push_back({ev[cursor], ev[cursor + 1], ev[cursor + 2]});
        cursor += 3;
    }
    const Fe h_eval = ev[h_slot];
    sw.lap(8);
    // 10. multiopen (SHPLONK)
    enum : uint32_t { K_ADV = 1, K_FIX, K_H, K_RND, K_SIGMA, K_Z, K_M, K_PHI };
    // halo2's query order -- advice, permutation products, lookups, fixed, sigma, h, random -- fixes the order of SHPLONK's rotation
    // sets (first appearance) and of the commitments inside each; pinned on the reference's generated EVM verifier
    // (tests/test_evm_verifier.py).  The verifier rebuilds the same list with commitments for polynomials.
    auto is_mine = [&](uint32_t own) { return own == topo.rank; };
    std::vector<OpenQuery> qs;
    for (auto& q : cs.advice_queries)
        qs.push_back({{K_ADV, q.col}, adv_polys[q.col], rot_point(q.rot), adv_evals[{q.col, q.rot}], is_mine(owner_of(adv_polys[q.col], adv_owner[q.col]))});
    for (uint32_t j = 0; j < z_polys.size(); j++) {
        const bool mn = is_mine(owner_of(z_polys[j], z_owner[j]));
        qs.push_back({{K_Z, j}, z_polys[j], x, z_evals[j].e0, mn});
        qs.push_back({{K_Z, j}, z_polys[j], rot_point(1), z_evals[j].e1, mn});
        if (z_evals[j].has2) qs.push_back({{K_Z, j}, z_polys[j], rot_point((int32_t)u), z_evals[j].e2, mn});
    }
    for (uint32_t i = 0; i < nl; i++) {
        const bool mn = is_mine(owner_of(phi_polys[i], lk_owner[i]));
        qs.push_back({{K_PHI, i}, phi_polys[i], x, lk_evals[i][0], mn});
        qs.push_back({{K_PHI, i}, phi_polys[i], rot_point(1), lk_evals[i][1], mn});
        qs.push_back({{K_M, i}, m_polys[i], x, lk_evals[i][2], mn});
    }
    for (auto& q : cs.fixed_queries) qs.push_back({{K_FIX, q.col}, pk.fixed_polys[q.col], rot_point(q.rot), fix_evals[{q.col, q.rot}], is_mine(fix_own[q.col])});
    for (uint32_t i = 0; i < pk.sigma_polys.size(); i++) qs.push_back({{K_SIGMA, i}, pk.sigma_polys[i], x, sigma_evals[i], is_mine(sig_own[i])});
    qs.push_back({{K_H}, hcomb, x, h_eval, is_mine(h_own)});
    qs.push_back({{K_RND}, rnd, x, random_eval, is_mine(rnd_own)});
    shplonk_prove(be, T, qs, n);
    sw.lap(9);
    sw.total();
    return T.proof();
}


// ------------------------------------------------------------------ verify_proof (the SAFE self-check, verify_proof_circuit)
// halo2's verifier for the proofs create_proof above emits (/root/reference/src/pfsys/mod.rs:557-590 verify_proof_circuit, and the
// CheckMode::SAFE branch of create_proof_circuit :470-480): replay the transcript, recompute the quotient identity at x from the
// claimed evaluations, and check the SHPLONK opening with one pairing product against the SRS's g2 / s_g2.  O(#columns) field and
// group operations on the host (pairing.hpp); no device work.
struct ProofReader {
    const uint8_t* p;
    size_t len, off = 0;
    std::vector<uint8_t> buf;
    void need(size_t m) {
        if (off + m > len) throw Error(EZKL_ERR_INVALID, "proof truncated");
    }
    void common_scalar(const Fe& s) {
        uint8_t b[32];
        to_be32(s.canonical(), b);
        buf.insert(buf.end(), b, b + 32);
    }
    bn::P1 read_point() {
        need(64);
        const U256 x = from_be32(p + off), y = from_be32(p + off + 32);
        if (cmp(x, FQ.p) >= 0 || cmp(y, FQ.p) >= 0) throw Error(EZKL_ERR_INVALID, "non-canonical point in the proof");
        buf.insert(buf.end(), p + off, p + off + 64);
        off += 64;
        bn::P1 r;
        r.x = bn::Fq{mont_mul(x, FQ.r2, FQ)};
        r.y = bn::Fq{mont_mul(y, FQ.r2, FQ)};
        r.inf = r.x.is_zero() && r.y.is_zero();
        if (!r.inf && !(r.y * r.y == r.x * r.x * r.x + bn::Fq::from_u64(3))) throw Error(EZKL_ERR_INVALID, "point not on the curve");
        return r;
    }
    Fe read_scalar() {
        need(32);
        const U256 c = from_be32(p + off);
        if (cmp(c, FR.p) >= 0) throw Error(EZKL_ERR_INVALID, "non-canonical scalar in the proof");
        buf.insert(buf.end(), p + off, p + off + 32);
        off += 32;
        return Fe::from_canonical(c);
    }
    Fe squeeze_challenge() {
        if (buf.size() == 32) buf.push_back(0x01);
        auto h = keccak256(buf.data(), buf.size());
        buf.assign(h.begin(), h.end());
        return Fe::from_canonical(reduce_fr(from_be32(h.data())));
    }
};

static bool verify_proof(ConstraintSystem& cs, const std::vector<G1>& fixed_commitments, const std::vector<G1>& sigma_commitments, const Fe& digest,
                         const bn::G2& g2, const bn::G2& s_g2, const uint8_t* proof, size_t proof_len, const void* const* instances,
                         const uint32_t* instance_lens) {
    using bn::P1;
    const uint32_t n = cs.n, k = cs.k, u = cs.usable;
    ProofReader T{proof, proof_len, 0, {}};
    T.common_scalar(digest);
    std::vector<std::vector<Fe>> inst(cs.n_instance);
    for (uint32_t i = 0; i < cs.n_instance; i++) {
        if (instance_lens[i] > u) return false;
        for (uint32_t j = 0; j < instance_lens[i]; j++) {
            U256 v;
            std::memcpy(v.data(), (const uint8_t*)instances[i] + 32 * j, 32);
            if (cmp(v, FR.p) >= 0) return false;
            inst[i].push_back(Fe{v});
            T.common_scalar(Fe{v});
        }
    }
    std::vector<P1> adv_c(cs.n_advice);
    std::vector<Fe> user_chal;
    for (uint32_t phase = 0; phase < 2; phase++) {
        bool any = false;
        for (uint32_t c = 0; c < cs.n_advice; c++)
            if (cs.advice_phase[c] == phase) { adv_c[c] = T.read_point(); any = true; }
        if (phase == 0 && any)
            for (uint32_t i = 0; i < cs.n_challenges; i++) user_chal.push_back(T.squeeze_challenge());
    }
    const size_t nl = cs.lookups.size();
    Fe theta = Fe::zero();
    std::vector<P1> m_c, phi_c, z_c, h_c;
    if (nl) {
        theta = T.squeeze_challenge();
        for (size_t i = 0; i < nl; i++) m_c.push_back(T.read_point());
    }
    const Fe beta = T.squeeze_challenge(), gamma = T.squeeze_challenge();
    for (uint32_t j = 0; j < cs.n_chunks; j++) z_c.push_back(T.read_point());
    for (size_t i = 0; i < nl; i++) phi_c.push_back(T.read_point());
    const P1 rnd_c = T.read_point();
    const Fe y = T.squeeze_challenge();
    for (uint32_t i = 0; i + 1 < cs.degree; i++) h_c.push_back(T.read_point());
    const Fe x = T.squeeze_challenge();
    std::map<std::pair<uint32_t, int32_t>, Fe> ev[3];        // advice, fixed, instance evaluations by (column, rotation)
    for (auto& q : cs.advice_queries) ev[0][{q.col, q.rot}] = T.read_scalar();
    for (auto& q : cs.fixed_queries) ev[1][{q.col, q.rot}] = T.read_scalar();
    const Fe random_eval = T.read_scalar();
    std::vector<Fe> sigma_ev;
    for (size_t i = 0; i < cs.perm.size(); i++) sigma_ev.push_back(T.read_scalar());
    struct ZE { Fe e0, e1, e2; bool has2; };
    std::vector<ZE> z_ev;
    for (uint32_t j = 0; j < cs.n_chunks; j++) {
        ZE z{T.read_scalar(), T.read_scalar(), Fe::zero(), j + 1 < cs.n_chunks};
        if (z.has2) z.e2 = T.read_scalar();
        z_ev.push_back(z);
    }
    struct LE { Fe phi, phi_next, m; };
    std::vector<LE> lk_ev;
    for (size_t i = 0; i < nl; i++) lk_ev.push_back(LE{T.read_scalar(), T.read_scalar(), T.read_scalar()});
    const Fe w = omega(k);
    auto rot_point = [&](int32_t r) { return x * w.pow((uint64_t)(r >= 0 ? (uint32_t)r % n : n - ((uint32_t)(-r) % n))); };
    const Fe xn = x.pow((uint64_t)n), zx = xn - Fe::one(), ninv = Fe::from_u64(n).inv();
    auto lagrange = [&](const Fe& at, const Fe& at_n_minus_1, uint32_t i) {   // l_i(at) = omega^i (at^n - 1) / (n (at - omega^i))
        const Fe wi = w.pow((uint64_t)i);
        return wi * at_n_minus_1 * ninv * (at - wi).inv();
    };
    for (auto& q : cs.instance_queries) {                     // instance columns are evaluated by the verifier from the public values
        const Fe z = rot_point(q.rot), zn1 = z.pow((uint64_t)n) - Fe::one();
        Fe acc = Fe::zero();
        for (size_t i = 0; i < inst[q.col].size(); i++) acc = acc + inst[q.col][i] * lagrange(z, zn1, (uint32_t)i);
        ev[2][{q.col, q.rot}] = acc;
    }
    const Fe l0 = lagrange(x, zx, 0), llast = lagrange(x, zx, u);
    Fe lblind = Fe::zero();
    for (uint32_t i = u; i < n; i++) lblind = lblind + lagrange(x, zx, i);
    const Fe lact = Fe::one() - lblind;
    std::vector<Fe> memo(cs.nodes.size());
    std::vector<uint8_t> have(cs.nodes.size(), 0);
    std::function<Fe(uint32_t)> evalx = [&](uint32_t id) -> Fe {
        if (have[id]) return memo[id];
        const Node& nd = cs.nodes[id];
        Fe r;
        switch (nd.op) {
        case N_CONST: r = nd.c; break;
        case N_ADV: case N_FIX: case N_INST: {
            auto it = ev[nd.op - N_ADV].find({nd.a, (int32_t)nd.b});
            if (it == ev[nd.op - N_ADV].end()) throw Error(EZKL_ERR_INVALID, "expression reads an unqueried cell");
            r = it->second;
            break;
        }
        case N_CHAL: r = user_chal.at(nd.a); break;
        case N_NEG: r = -evalx(nd.a); break;
        case N_ADD: r = evalx(nd.a) + evalx(nd.b); break;
        case N_SUB: r = evalx(nd.a) - evalx(nd.b); break;
        default: r = evalx(nd.a) * evalx(nd.b); break;
        }
        have[id] = 1;
        return memo[id] = r;
    };
    std::vector<Fe> terms;
    for (uint32_t g : cs.gates) terms.push_back(evalx(g));
    if (!cs.perm.empty()) {
        terms.push_back(l0 * (Fe::one() - z_ev[0].e0));
        const Fe zl = z_ev.back().e0;
        terms.push_back(llast * (zl * zl - zl));
        for (uint32_t j = 1; j < cs.n_chunks; j++) terms.push_back(l0 * (z_ev[j].e0 - z_ev[j - 1].e2));
        uint32_t pos = 0;
        const Fe delta{FR_DELTA};
        Fe dpow = Fe::one();
        uint32_t j = 0;
        for (auto& chunk : cs.perm_chunks()) {
            Fe left = z_ev[j].e1, right = z_ev[j].e0;
            for (size_t i = 0; i < chunk.size(); i++) {
                const Fe v = ev[chunk[i].first - N_ADV].at({chunk[i].second, 0});
                left = left * (v + beta * sigma_ev[pos + i] + gamma);
                right = right * (v + beta * dpow * x + gamma);
                dpow = dpow * delta;
            }
            terms.push_back(lact * (left - right));
            pos += (uint32_t)chunk.size();
            j++;
        }
    }
    auto compress = [&](const std::vector<uint32_t>& tup) {
        Fe acc = evalx(tup[0]);
        for (size_t i = 1; i < tup.size(); i++) acc = acc * theta + evalx(tup[i]);
        return acc;
    };
    for (size_t li = 0; li < nl; li++) {
        const Lookup& l = cs.lookups[li];
        std::vector<Fe> fb;
        for (auto& t : l.inputs) fb.push_back(compress(t) + beta);
        const Fe tb = compress(l.table) + beta;
        Fe prodf = Fe::one(), ssum = Fe::zero();
        for (auto& f : fb) prodf = prodf * f;
        for (size_t jj = 0; jj < fb.size(); jj++) {
            Fe pj = Fe::one();
            for (size_t i2 = 0; i2 < fb.size(); i2++)
                if (i2 != jj) pj = pj * fb[i2];
            ssum = ssum + pj;
        }
        const Fe lhs = (lk_ev[li].phi_next - lk_ev[li].phi) * prodf * tb, rhs = ssum * tb - lk_ev[li].m * prodf;
        terms.push_back(l0 * lk_ev[li].phi);
        terms.push_back(llast * lk_ev[li].phi);
        terms.push_back(lact * (lhs - rhs));
    }
    Fe num = Fe::zero();
    for (auto& t : terms) num = num * y + t;
    const Fe h_eval = num * zx.inv();
    // ---- the opening queries, in the prover's order: (commitment, point, evaluation) grouped like shplonk_prove
    P1 hc{};
    for (size_t i = h_c.size(); i-- > 0;) hc = bn::p1_add(bn::p1_mul(hc, xn), h_c[i]);
    struct VQ { std::vector<uint32_t> key; P1 com; Fe point, eval; };
    enum : uint32_t { K_ADV = 1, K_FIX, K_H, K_RND, K_SIGMA, K_Z, K_M, K_PHI };
    std::vector<VQ> qs;
    for (auto& q : cs.advice_queries) qs.push_back({{K_ADV, q.col}, adv_c[q.col], rot_point(q.rot), ev[0][{q.col, q.rot}]});
    for (uint32_t j = 0; j < z_ev.size(); j++) {
        qs.push_back({{K_Z, j}, z_c[j], x, z_ev[j].e0});
        qs.push_back({{K_Z, j}, z_c[j], rot_point(1), z_ev[j].e1});
        if (z_ev[j].has2) qs.push_back({{K_Z, j}, z_c[j], rot_point((int32_t)u), z_ev[j].e2});
    }
    for (uint32_t i = 0; i < nl; i++) {
        qs.push_back({{K_PHI, i}, phi_c[i], x, lk_ev[i].phi});
        qs.push_back({{K_PHI, i}, phi_c[i], rot_point(1), lk_ev[i].phi_next});
        qs.push_back({{K_M, i}, m_c[i], x, lk_ev[i].m});
    }
    for (auto& q : cs.fixed_queries) qs.push_back({{K_FIX, q.col}, bn::p1_from(fixed_commitments[q.col]), rot_point(q.rot), ev[1][{q.col, q.rot}]});
    for (uint32_t i = 0; i < sigma_ev.size(); i++) qs.push_back({{K_SIGMA, i}, bn::p1_from(sigma_commitments[i]), x, sigma_ev[i]});
    qs.push_back({{K_H}, hc, x, h_eval});
    qs.push_back({{K_RND}, rnd_c, x, random_eval});
    struct VPoly { P1 com; std::map<U256, std::pair<Fe, Fe>> ev; };
    std::vector<VPoly> polys;
    std::map<std::vector<uint32_t>, size_t> by_key;
    for (auto& q : qs) {
        auto it = by_key.find(q.key);
        if (it == by_key.end()) {
            it = by_key.emplace(q.key, polys.size()).first;
            polys.push_back(VPoly{q.com, {}});
        }
        polys[it->second].ev[q.point.canonical()] = {q.point, q.eval};
    }
    struct VGroup { std::vector<U256> pts; std::vector<Fe> pts_fe; std::vector<size_t> members; };
    std::vector<VGroup> groups;
    for (size_t i = 0; i < polys.size(); i++) {
        std::vector<U256> pts;
        for (auto& e : polys[i].ev) pts.push_back(e.first);
        std::sort(pts.begin(), pts.end(), u256_less);
        size_t gi = 0;
        for (; gi < groups.size(); gi++)
            if (groups[gi].pts == pts) break;
        if (gi == groups.size()) {
            VGroup gn;
            gn.pts = pts;
            for (auto& p : pts) gn.pts_fe.push_back(polys[i].ev[p].first);
            groups.push_back(gn);
        }
        groups[gi].members.push_back(i);
    }
    const Fe ys = T.squeeze_challenge();
    std::vector<U256> all_pts;
    std::map<U256, Fe> pt_fe;
    for (auto& gr : groups)
        for (size_t i = 0; i < gr.pts.size(); i++) { all_pts.push_back(gr.pts[i]); pt_fe[gr.pts[i]] = gr.pts_fe[i]; }
    std::sort(all_pts.begin(), all_pts.end(), u256_less);
    all_pts.erase(std::unique(all_pts.begin(), all_pts.end()), all_pts.end());
    const Fe v = T.squeeze_challenge();
    const P1 pi1 = T.read_point();
    const Fe uu = T.squeeze_challenge();
    const P1 pi2 = T.read_point();
    if (T.off != proof_len) return false;
    Fe zt_u = Fe::one();
    for (auto& p : all_pts) zt_u = zt_u * (uu - pt_fe[p]);
    P1 gen;
    gen.inf = false; gen.x = bn::Fq::one(); gen.y = bn::Fq::from_u64(2);
    P1 L{};
    Fe pw = Fe::one(), norm = Fe::one();
    bool first = true;
    for (auto& gr : groups) {
        P1 qc{};
        std::vector<Fe> evs(gr.pts.size(), Fe::zero());
        Fe yp = Fe::one();
        for (size_t mi : gr.members) {
            qc = bn::p1_add(qc, bn::p1_mul(polys[mi].com, yp));
            for (size_t i = 0; i < gr.pts.size(); i++) evs[i] = evs[i] + yp * polys[mi].ev[gr.pts[i]].second;
            yp = yp * ys;
        }
        const std::vector<Fe> r = interpolate(gr.pts_fe, evs);
        Fe zdiff = Fe::one();
        for (auto& p : all_pts)
            if (!std::binary_search(gr.pts.begin(), gr.pts.end(), p, u256_less)) zdiff = zdiff * (uu - pt_fe[p]);
        if (first) { norm = zdiff.inv(); first = false; }    // halo2: coefficients normalised by the first set's
        const P1 term = bn::p1_add(qc, bn::p1_neg(bn::p1_mul(gen, eval_small(r, uu))));
        L = bn::p1_add(L, bn::p1_mul(term, pw * zdiff * norm));
        pw = pw * v;
    }
    L = bn::p1_add(L, bn::p1_neg(bn::p1_mul(pi1, zt_u * norm)));
    const P1 lhs = bn::p1_add(L, bn::p1_mul(pi2, uu));
    return bn::pairing_check({{pi2, s_g2}, {bn::p1_neg(lhs), g2}});
}

}  // namespace ezkl_prover

// ------------------------------------------------------------------ C ABI
using namespace ezkl_prover;
struct ezkl_prover_cs {
    std::unique_ptr<ConstraintSystem> cs;
};
struct ezkl_prover_pk {
    std::unique_ptr<ProvingKey> pk;
};
template <class F>
static int guarded(F&& f) {
    try {
        g_last_error.clear();
        f();
        return EZKL_OK;
    } catch (const Error& e) {
        g_last_error = e.what();
        return e.code;
    } catch (const std::bad_alloc&) {
        g_last_error = "host allocation failed";
        return EZKL_ERR_NOMEM;
    } catch (const std::exception& e) {
        g_last_error = e.what();
        return EZKL_ERR_INVALID;
    }
}
extern "C" {
int ezkl_prover_cs_parse(const void* blob, size_t len, ezkl_cs_t* out) {
    if (!blob || !out) return EZKL_ERR_INVALID;
    return guarded([&] { *out = new ezkl_prover_cs{parse_cs(blob, len)}; });
}
int ezkl_prover_cs_free(ezkl_cs_t cs) {
    delete cs;
    return EZKL_OK;
}
int ezkl_prover_cs_info(ezkl_cs_t h, uint32_t out[8]) {
    if (!h || !out) return EZKL_ERR_INVALID;
    const ConstraintSystem& cs = *h->cs;
    const uint32_t v[8] = {cs.degree, cs.ext_k, cs.chunk, cs.n_chunks, cs.usable, (uint32_t)cs.advice_queries.size(), (uint32_t)cs.fixed_queries.size(),
                           (uint32_t)cs.instance_queries.size()};
    std::memcpy(out, v, sizeof v);
    return EZKL_OK;
}
int ezkl_prover_cs_set_shard(ezkl_cs_t h, uint32_t lo, uint32_t hi, ezkl_fold_fn fold, void* user) {
    if (!h) return EZKL_ERR_INVALID;
    if (hi == 0 && lo == 0) {                  // back to one GPU
        h->cs->shard = Shard();
        return EZKL_OK;
    }
    if (!fold || lo >= hi || hi > h->cs->n) return EZKL_ERR_INVALID;
    Shard s = h->cs->shard;                    // keeps a sweep gather installed earlier
    s.lo = lo; s.hi = hi; s.fold = fold; s.user = user;
    h->cs->shard = s;
    return EZKL_OK;
}
// the same sharding with the library's own RCCL communicator (include/ezkl_hip.h ezkl_hip_comm_*): no caller callbacks, no torch
static int comm_fold(void*, void* points, uint32_t count) { return ezkl_hip_comm_fold_points(points, count); }
static int comm_gather(void*, void* buf, size_t total, size_t, size_t) { return ezkl_hip_comm_allgather_dev(buf, total); }
static int comm_allgather_host(void*, void* buf, size_t per) { return ezkl_hip_comm_allgather_host(buf, per); }
// the sweep exchange of the column-sharded prover: ezkl_hip_comm_alltoallv_dev (grouped ncclSend / ncclRecv over xGMI, all peers at once)
static int comm_alltoall(void*, const ezkl_comm_seg_t* sends, size_t n_sends, const ezkl_comm_seg_t* recvs, size_t n_recvs) {
    return ezkl_hip_comm_alltoallv_dev(sends, n_sends, recvs, n_recvs);
}
int ezkl_prover_cs_set_shard_comm(ezkl_cs_t h) {
    if (!h) return EZKL_ERR_INVALID;
    int world = 0, rank = 0;
    int rc = ezkl_hip_comm_info(&world, &rank);
    if (rc) return rc;
    if (world < 1) return EZKL_ERR_INVALID;               // ezkl_hip_comm_init first
    const uint32_t n = h->cs->n;
    const uint32_t base = n / (uint32_t)world, rem = n % (uint32_t)world;
    const uint32_t lo = (uint32_t)rank * base + std::min((uint32_t)rank, rem), hi = lo + base + ((uint32_t)rank < rem ? 1 : 0);
    rc = ezkl_prover_cs_set_shard(h, lo, hi, comm_fold, nullptr);
    if (rc) return rc;
    if (world > 1 && (world & (world - 1)) == 0 && n % (uint32_t)world == 0) {
        rc = ezkl_prover_cs_set_sweep_gather(h, comm_gather, nullptr);
        if (rc) return rc;
        // columns and arguments by owner (takes effect once the caller declares complete base sets: ezkl_prover_cs_set_shard_full_bases)
        return ezkl_prover_cs_set_shard_exchange(h, comm_allgather_host, comm_alltoall, nullptr);
    }
    if (world > 1) {
        // Columns and arguments by owner, and the row-sharded sweep, divide the 2^k rows and the 2^(ext_k - k) cosets into equal parts: they
        // need a power-of-two number of ranks that divides 2^k.  Any other world still proves -- every commitment sharded by point ranges and
        // folded, everything else replicated on every rank (round 1's mode) -- and says so once instead of silently being slower.
        static bool said = false;
        if (!said) {
            said = true;
            fprintf(stderr, "[ezkl_prover] %d ranks: the owner mode (columns / arguments by owner, row-sharded sweep) needs a power-of-two number of ranks; "
                            "this run shards the commitments by point ranges and replicates the rest\n", world);
        }
    }
    return EZKL_OK;
}
int ezkl_prover_cs_set_shard_exchange(ezkl_cs_t h, ezkl_allgather_host_fn allgather_host, ezkl_exchange_fn exchange, void* user) {
    if (!h || ((allgather_host == nullptr) != (exchange == nullptr))) return EZKL_ERR_INVALID;
    h->cs->shard.allgather_host = allgather_host;
    h->cs->shard.exchange = exchange;
    h->cs->shard.xuser = user;
    return EZKL_OK;
}
int ezkl_prover_cs_shard_stats(ezkl_cs_t h, uint64_t out[4]) {
    if (!h || !out) return EZKL_ERR_INVALID;
    for (int i = 0; i < 4; i++) out[i] = h->cs->shard.stats[i];
    return EZKL_OK;
}
int ezkl_prover_cs_set_shard_full_bases(ezkl_cs_t h, int on) {
    if (!h) return EZKL_ERR_INVALID;
    h->cs->shard.full_bases = on != 0;
    return EZKL_OK;
}
int ezkl_prover_cs_set_sweep_gather(ezkl_cs_t h, ezkl_gather_fn gather, void* user) {
    if (!h) return EZKL_ERR_INVALID;
    h->cs->shard.gather = gather;
    h->cs->shard.gather_user = user;
    return EZKL_OK;
}
int ezkl_prover_cs_set_advice_by_pointer(ezkl_cs_t h, int on) {
    if (!h) return EZKL_ERR_INVALID;
    h->cs->advice_by_pointer = on != 0;
    return EZKL_OK;
}
int ezkl_prover_cs_sharded_sweeps(ezkl_cs_t h, uint64_t* out) {
    if (!h || !out) return EZKL_ERR_INVALID;
    *out = h->cs->shard.sharded_sweeps;
    return EZKL_OK;
}
int ezkl_prover_keygen(ezkl_cs_t cs, ezkl_bases_t g, const void* const* fixed_values, const uint32_t* copies, size_t n_copies, ezkl_pk_t* out) {
    if (!cs || !g || !out || (cs->cs->n_fixed && !fixed_values) || (n_copies && !copies)) return EZKL_ERR_INVALID;
    return guarded([&] {
        invalid(ezkl_hip_bases_len(g) < (cs->cs->shard.on() ? cs->cs->shard.hi - cs->cs->shard.lo : cs->cs->n), "SRS smaller than 2^k (or than this rank's slice)");
        *out = new ezkl_prover_pk{keygen(*cs->cs, g, fixed_values, copies, n_copies)};
        const auto t0 = std::chrono::steady_clock::now();
        prepare_quotient(*(*out)->pk);      // `setup` pays hiprtc for the circuit's sweep kernel (cached on disk): the first `prove` does not
        if (getenv("EZKL_PROVER_KEYGEN_TIMING"))
            fprintf(stderr, "[ezkl_prover] keygen %-36s %8.1f ms\n", "sweep kernels (hiprtc or disk cache)",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    });
}
// which cosets of the extended domain the key's extended columns hold (owner mode: the ones this rank sweeps) and how many bytes of HBM
// the key occupies: out[0] = first coset, out[1] = count, out[2] = E, out[3] = resident key bytes
int ezkl_prover_pk_residency(ezkl_pk_t pk, uint64_t out[4]) {
    if (!pk || !out) return EZKL_ERR_INVALID;
    const ProvingKey& k_ = *pk->pk;
    const ConstraintSystem& cs = *k_.cs;
    const uint64_t E = 1ull << (cs.ext_k - cs.k), n = cs.n;
    out[0] = k_.coset_first; out[1] = k_.stream ? 0 : k_.coset_count; out[2] = E;        // a streamed key holds NO coset of its fixed / permutation columns
    const uint64_t small = (uint64_t)(k_.fixed_values.size() + k_.fixed_polys.size() + k_.sigma_values.size() + k_.sigma_polys.size() + 1) * n * 32;
    const uint64_t ext = (uint64_t)((k_.stream ? 0 : k_.fixed_cosets.size() + k_.sigma_cosets.size()) + 4) * k_.coset_count * n * 32;
    out[3] = small + ext;
    return EZKL_OK;
}
int ezkl_prover_pk_sweep_stats(ezkl_pk_t pk, uint64_t out[4]) {
    if (!pk || !out) return EZKL_ERR_INVALID;
    return guarded([&] { sweep_stats(*pk->pk, out); });
}
int ezkl_prover_pk_write(ezkl_pk_t pk, void* out, size_t cap, size_t* len) {
    if (!pk || !len) return EZKL_ERR_INVALID;
    return guarded([&] {
        std::vector<uint8_t> b = pk_write(*pk->pk);
        *len = b.size();
        if (b.size() > cap || !out) throw Error(EZKL_ERR_NOMEM, "key buffer too small");
        std::memcpy(out, b.data(), b.size());
    });
}
int ezkl_prover_pk_read(ezkl_cs_t cs, const void* buf, size_t len, ezkl_pk_t* out) {
    if (!cs || !buf || !out) return EZKL_ERR_INVALID;
    return guarded([&] { *out = new ezkl_prover_pk{pk_read(*cs->cs, (const uint8_t*)buf, len)}; });
}
int ezkl_prover_pk_read_file(ezkl_cs_t cs, const char* path, ezkl_pk_t* out) {
    if (!cs || !path || !out) return EZKL_ERR_INVALID;
    return guarded([&] { *out = new ezkl_prover_pk{pk_read_file(*cs->cs, path)}; });
}
int ezkl_prover_pk_set_selectors(ezkl_pk_t pk, const void* bits, size_t len) {
    return guarded([&] {
        if (!pk || (!bits && len)) throw Error(EZKL_ERR_INVALID, "null handle");
        const ConstraintSystem& cs = *pk->pk->cs;
        if (len != (size_t)cs.n_selectors * ((cs.n + 7) / 8)) throw Error(EZKL_ERR_INVALID, "selector section of unexpected length");
        pk->pk->selector_bits.assign((const uint8_t*)bits, (const uint8_t*)bits + len);
    });
}
int ezkl_prover_pk_recommit(ezkl_pk_t pk, ezkl_bases_t g) {
    return guarded([&] {
        if (!pk || !g) throw Error(EZKL_ERR_INVALID, "null handle");
        pk_recommit(*pk->pk, g);
    });
}
int ezkl_prover_pk_free(ezkl_pk_t pk) {
    delete pk;
    return EZKL_OK;
}
int ezkl_prover_vk(ezkl_pk_t h, void* fixed_commitments, void* permutation_commitments, void* digest) {
    if (!h) return EZKL_ERR_INVALID;
    const ProvingKey& pk = *h->pk;
    if (fixed_commitments && !pk.fixed_commitments.empty()) std::memcpy(fixed_commitments, pk.fixed_commitments.data(), pk.fixed_commitments.size() * 64);
    if (permutation_commitments && !pk.sigma_commitments.empty()) std::memcpy(permutation_commitments, pk.sigma_commitments.data(), pk.sigma_commitments.size() * 64);
    if (digest) std::memcpy(digest, pk.digest.v.data(), 32);
    return EZKL_OK;
}
int ezkl_prover_pk_set_transcript_repr(ezkl_pk_t h, const void* repr) {
    if (!h || !repr) return EZKL_ERR_INVALID;
    U256 v;
    std::memcpy(v.data(), repr, 32);
    if (cmp(v, FR.p) >= 0) return EZKL_ERR_INVALID;              // a canonical scalar, as halo2 transcript_repr is
    h->pk->digest = Fe::from_canonical(v);
    return EZKL_OK;
}
int ezkl_prover_create_proof(ezkl_pk_t pk, ezkl_bases_t g, ezkl_bases_t g_lagrange, const void* const* advice, ezkl_advice_fn advice_fn, void* advice_user,
                             const void* const* instances, const uint32_t* instance_lens, ezkl_rng_fn rng, void* rng_user, uint64_t seed, void* proof_out,
                             size_t cap, size_t* proof_len, double* timings) {
    return ezkl_prover_create_proof_fmt(pk, g, g_lagrange, advice, nullptr, advice_fn, advice_user, instances, instance_lens, rng, rng_user, seed, proof_out, cap,
                                        proof_len, timings);
}
int ezkl_prover_create_proof_fmt(ezkl_pk_t pk, ezkl_bases_t g, ezkl_bases_t g_lagrange, const void* const* advice, const uint8_t* advice_formats,
                                 ezkl_advice_fn advice_fn, void* advice_user, const void* const* instances, const uint32_t* instance_lens, ezkl_rng_fn rng,
                                 void* rng_user, uint64_t seed, void* proof_out, size_t cap, size_t* proof_len, double* timings) {
    if (!pk || !g || !g_lagrange || !proof_len) return EZKL_ERR_INVALID;
    if (pk->pk->cs->n_instance && (!instances || !instance_lens)) return EZKL_ERR_INVALID;
    return guarded([&] {
        const Shard& sh = pk->pk->cs->shard;
        const size_t want = sh.on() && !sh.full_bases ? sh.hi - sh.lo : pk->pk->cs->n;
        invalid(ezkl_hip_bases_len(g) < want || ezkl_hip_bases_len(g_lagrange) != want, "SRS size does not match 2^k (or this rank's slice)");
        Rng r(rng, rng_user, seed);
        if (sh.on() && !rng && seed == 0) {
            // every rank must blind with the SAME randomness to emit the same proof: rank 0's 256-bit OS-entropy key goes to everyone over
            // the library communicator (full entropy, fresh per proof -- not a 64-bit seed).  Callback-sharded provers without the
            // communicator must pass an rng callback that is identical on every rank, or a det-prove seed.
            int world = 0, rank = 0;
            check(ezkl_hip_comm_info(&world, &rank), "ezkl_hip_comm_info");
            invalid(world < 1, "sharded proving needs the same randomness on every rank: pass a seed or an rng callback, or shard over ezkl_hip_comm_init");
            check(ezkl_hip_comm_broadcast_host(r.key, sizeof r.key, 0), "ezkl_hip_comm_broadcast_host");
        }
        std::vector<uint8_t> proof = create_proof(*pk->pk, g, g_lagrange, advice, advice_fn, advice_user, instances, instance_lens, r, timings, advice_formats);
        *proof_len = proof.size();
        if (proof.size() > cap || !proof_out) throw Error(EZKL_ERR_NOMEM, "proof buffer too small");
        std::memcpy(proof_out, proof.data(), proof.size());
    });
}
int ezkl_prover_verify_proof(ezkl_pk_t pk, const void* g2, const void* s_g2, const void* proof, size_t proof_len, const void* const* instances,
                             const uint32_t* instance_lens, int* accepted) {
    if (!pk || !g2 || !s_g2 || !proof || !accepted) return EZKL_ERR_INVALID;
    *accepted = 0;
    if (pk->pk->cs->n_instance && (!instances || !instance_lens)) return EZKL_ERR_INVALID;
    return guarded([&] {
        const ProvingKey& k = *pk->pk;
        const bn::G2 a = bn::g2_from_bytes((const uint8_t*)g2), b = bn::g2_from_bytes((const uint8_t*)s_g2);
        if (!bn::g2_on_curve(a) || !bn::g2_on_curve(b) || a.inf || b.inf) throw Error(EZKL_ERR_INVALID, "g2 / s_g2 not on the twist");
        bool ok = false;
        try {
            ok = verify_proof(*k.cs, k.fixed_commitments, k.sigma_commitments, k.digest, a, b, (const uint8_t*)proof, proof_len, instances, instance_lens);
        } catch (const Error& e) {             // a malformed proof is a rejection, not a failure of the call
            if (e.code != EZKL_ERR_INVALID) throw;
            g_last_error = e.what();
        }
        *accepted = ok ? 1 : 0;
    });
}
// verify_proof from the verifying key ALONE (the reference's `verify` reads settings + vk.key, /root/reference/src/execute.rs:1651): the vk
// file is halo2's raw-bytes layout [3, k, compress] | u32 LE #fixed | #fixed x G1 | #perm x G1 | selector bits; the constraint system
// supplies the counts.  Host only: no device, no proving key, no private data.
int ezkl_prover_verify_proof_vk(ezkl_cs_t cs, const void* vk_buf, size_t vk_len, const void* g2, const void* s_g2, const void* proof, size_t proof_len,
                                const void* const* instances, const uint32_t* instance_lens, int* accepted) {
    if (!cs || !vk_buf || !g2 || !s_g2 || !proof || !accepted) return EZKL_ERR_INVALID;
    *accepted = 0;
    if (cs->cs->n_instance && (!instances || !instance_lens)) return EZKL_ERR_INVALID;
    return guarded([&] {
        ConstraintSystem& c = *cs->cs;
        const uint8_t* b = (const uint8_t*)vk_buf;
        const size_t np = c.perm.size(), want = 7 + 64 * ((size_t)c.n_fixed + np);
        invalid(vk_len < want, "verifying key truncated");
        invalid(b[0] != 3, "unsupported key version");
        invalid(b[1] != c.k, "key was made for another k");
        uint32_t nf = 0;
        for (int i = 0; i < 4; i++) nf |= (uint32_t)b[3 + i] << (8 * i);
        invalid(nf != c.n_fixed, "key has another number of fixed columns");
        ProvingKey vk;                                   // only the verifying-key part is filled
        vk.cs = &c;
        vk.fixed_commitments.resize(nf);
        vk.sigma_commitments.resize(np);
        if (nf) std::memcpy(vk.fixed_commitments.data(), b + 7, 64 * (size_t)nf);
        if (np) std::memcpy(vk.sigma_commitments.data(), b + 7 + 64 * (size_t)nf, 64 * np);
        // a damaged or foreign key is rejected here, not fed to the pairing arithmetic: every commitment must be the identity
        // encoding (0, 0) or a canonical point of the curve, and what follows the commitments must be a complete selector section (a vk.key
        // ends there; a pk.key -- whose prefix the vk is -- goes on with the polynomials)
        auto check_point = [&](const G1& p) {
            invalid(cmp(p.x, FQ.p) >= 0 || cmp(p.y, FQ.p) >= 0, "non-canonical coordinate in the verifying key");
            const bn::Fq x{p.x}, y{p.y};
            if (x.is_zero() && y.is_zero()) return;
            invalid(!(y * y == x * x * x + bn::Fq::from_u64(3)), "a commitment of the verifying key is not on the curve");
        };
        for (auto& p_ : vk.fixed_commitments) check_point(p_);
        for (auto& p_ : vk.sigma_commitments) check_point(p_);
        const size_t sel_bytes = (size_t)c.n_selectors * ((c.n + 7) / 8);
        invalid(vk_len < want + sel_bytes, "verifying key truncated (selector section)");
        const Fe digest = vk_digest(vk);
        const bn::G2 a = bn::g2_from_bytes((const uint8_t*)g2), bb = bn::g2_from_bytes((const uint8_t*)s_g2);
        if (!bn::g2_on_curve(a) || !bn::g2_on_curve(bb) || a.inf || bb.inf) throw Error(EZKL_ERR_INVALID, "g2 / s_g2 not on the twist");
        bool ok = false;
        try {
            ok = verify_proof(c, vk.fixed_commitments, vk.sigma_commitments, digest, a, bb, (const uint8_t*)proof, proof_len, instances, instance_lens);
        } catch (const Error& e) {
            if (e.code != EZKL_ERR_INVALID) throw;
            g_last_error = e.what();
        }
        *accepted = ok ? 1 : 0;
    });
}
int ezkl_prover_g2_mul_generator(const void* scalar, void* out128) {
    if (!scalar || !out128) return EZKL_ERR_INVALID;
    return guarded([&] {
        U256 sv;
        std::memcpy(sv.data(), scalar, 32);
        if (cmp(sv, FR.p) >= 0) throw Error(EZKL_ERR_INVALID, "non-canonical scalar");
        bn::g2_to_bytes(bn::g2_mul(bn::g2_generator(), Fe{sv}.canonical()), (uint8_t*)out128);
    });
}
int ezkl_prover_keccak256(const void* data, size_t len, void* out32) {
    if ((!data && len) || !out32) return EZKL_ERR_INVALID;
    auto h = keccak256((const uint8_t*)data, len);
    std::memcpy(out32, h.data(), 32);
    return EZKL_OK;
}
const char* ezkl_prover_last_error(void) { return g_last_error.c_str(); }
}


// ------------------------------------------------------------------ one process, several GPUs: the prover group
// `ezkl prove` is ONE process (/root/reference/src/execute.rs:1575-1627).  A group is the owner-mode prover above run by N host threads
// of that process, thread r bound to context r of libezkl_hip.so (ezkl_hip_set_context; normally one context per device,
// ezkl_hip_init(-1)).  The collectives are in-process: commitments are folded and scalars gathered through shared host memory behind a
// barrier, h is all_gathered and the sweep's row slabs are exchanged by peer copies (ezkl_hip_memcpy_peer, every thread PULLING what
// it needs into its own device memory).  No RCCL, no second process, no shared-memory hand-off of the witness: every thread reads
// the caller's advice columns in place.
namespace ezkl_prover {
struct ThreadComm {
    int world = 1;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    uint64_t generation = 0;
    bool failed = false;
    std::vector<std::vector<uint8_t>> host;                 // per rank: what it contributes to a fold / host all_gather
    std::vector<void*> dev_ptr;                             // per rank: the device buffer of an in-place all_gather
    std::vector<const ezkl_comm_seg_t*> sends;              // per rank: its send list of the current exchange
    std::vector<size_t> n_sends;
    explicit ThreadComm(int w) : world(w), host(w), dev_ptr(w, nullptr), sends(w, nullptr), n_sends(w, 0) {}
    // all threads of the group arrive, or (after a failure anywhere) nobody waits for the missing ones
    bool barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (failed) return false;
        const uint64_t gen = generation;
        if (++waiting == world) {
            waiting = 0;
            generation++;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen || failed; });
        }
        return !failed;
    }
    void fail() {
        std::lock_guard<std::mutex> lk(mu);
        failed = true;
        cv.notify_all();
    }
};
struct ThreadRank {
    ThreadComm* comm;
    int rank;
};
static int tc_fold(void* user, void* points, uint32_t count) {
    ThreadRank* tr = (ThreadRank*)user;
    ThreadComm& C = *tr->comm;
    C.host[tr->rank].assign((const uint8_t*)points, (const uint8_t*)points + 64 * (size_t)count);
    if (!C.barrier()) return 1;
    uint8_t* out = (uint8_t*)points;
    for (uint32_t j = 0; j < count; j++) {
        uint8_t acc[64];
        std::memcpy(acc, C.host[0].data() + 64 * (size_t)j, 64);
        for (int r = 1; r < C.world; r++) (void)ezkl_hip_g1_add_affine(acc, C.host[r].data() + 64 * (size_t)j, acc);
        std::memcpy(out + 64 * (size_t)j, acc, 64);
    }
    return C.barrier() ? 0 : 1;                              // nobody overwrites its contribution while others still read it
}
static int tc_allgather_host(void* user, void* buf, size_t per) {
    ThreadRank* tr = (ThreadRank*)user;
    ThreadComm& C = *tr->comm;
    const uint8_t* mine = (const uint8_t*)buf + per * (size_t)tr->rank;
    C.host[tr->rank].assign(mine, mine + per);
    if (!C.barrier()) return 1;
    for (int r = 0; r < C.world; r++)
        if (r != tr->rank) std::memcpy((uint8_t*)buf + per * (size_t)r, C.host[r].data(), per);
    return C.barrier() ? 0 : 1;
}
static int tc_gather(void* user, void* buf_dev, size_t total, size_t, size_t bytes) {
    ThreadRank* tr = (ThreadRank*)user;
    ThreadComm& C = *tr->comm;
    if (total != bytes * (size_t)C.world) return 1;
    if (ezkl_hip_synchronize() != EZKL_OK) { C.fail(); return 1; }          // my slice is complete before anybody pulls it
    C.dev_ptr[tr->rank] = buf_dev;
    if (!C.barrier()) return 1;
    for (int r = 0; r < C.world; r++) {
        if (r == tr->rank) continue;
        if (ezkl_hip_memcpy_peer((uint8_t*)buf_dev + bytes * (size_t)r, tr->rank, (const uint8_t*)C.dev_ptr[r] + bytes * (size_t)r, r, bytes) != EZKL_OK) {
            C.fail();
            return 1;
        }
    }
    if (ezkl_hip_synchronize() != EZKL_OK) { C.fail(); return 1; }          // my pulls are done: the others may go on writing their buffers
    return C.barrier() ? 0 : 1;
}
static int tc_exchange(void* user, const ezkl_comm_seg_t* sends, size_t n_sends, const ezkl_comm_seg_t* recvs, size_t n_recvs) {
    ThreadRank* tr = (ThreadRank*)user;
    ThreadComm& C = *tr->comm;
    if (ezkl_hip_synchronize() != EZKL_OK) { C.fail(); return 1; }
    C.sends[tr->rank] = sends;
    C.n_sends[tr->rank] = n_sends;
    if (!C.barrier()) return 1;
    // pull: my j-th segment from peer p is p's k-th segment to me, k = how many of my earlier receives came from p
    std::vector<size_t> cursor(C.world, 0);
    bool ok = true;
    for (size_t j = 0; j < n_recvs && ok; j++) {
        const int p = recvs[j].peer;
        size_t& k = cursor[p];
        while (k < C.n_sends[p] && C.sends[p][k].peer != tr->rank) k++;
        if (k == C.n_sends[p] || C.sends[p][k].bytes != recvs[j].bytes) { ok = false; break; }
        ok = ezkl_hip_memcpy_peer(recvs[j].ptr, tr->rank, C.sends[p][k].ptr, p, recvs[j].bytes) == EZKL_OK;
        k++;
    }
    if (ok) ok = ezkl_hip_synchronize() == EZKL_OK;         // the peer copies are stream-ordered: done before the sources may change
    if (!ok) { C.fail(); return 1; }
    return C.barrier() ? 0 : 1;                              // the send lists stay valid until everybody has pulled
}
}  // namespace ezkl_prover

struct ezkl_prover_group {
    int world = 1;
    std::vector<ezkl_cs_t> cs;
    std::vector<ezkl_bases_t> g, gl;
    std::vector<ezkl_pk_t> pk;
    std::unique_ptr<ThreadComm> comm;
    std::vector<ThreadRank> ranks;
    std::string error;
    // run f(rank) on every context's thread; the first error wins
    template <class F>
    int run(F&& f) {
        std::vector<int> rc(world, EZKL_OK);
        std::vector<std::string> msg(world);
        std::vector<std::thread> th;
        {
            std::lock_guard<std::mutex> lk(comm->mu);        // a fresh run after a failed one
            comm->failed = false;
            comm->waiting = 0;
        }
        for (int r = 0; r < world; r++)
            th.emplace_back([&, r] {
                rc[r] = ezkl_hip_set_context(r);
                if (rc[r] == EZKL_OK) rc[r] = f(r);
                if (rc[r] != EZKL_OK) {
                    msg[r] = ezkl_prover_last_error();
                    comm->fail();
                }
            });
        for (auto& t : th) t.join();
        for (int r = 0; r < world; r++)
            if (rc[r] != EZKL_OK) {
                error = "context " + std::to_string(r) + ": " + msg[r];
                g_last_error = error;
                return rc[r];
            }
        return EZKL_OK;
    }
};
extern "C" {
int ezkl_prover_group_create(const void* blob, size_t len, int n_contexts, ezkl_group_t* out) {
    if (!blob || !out || n_contexts < 1) return EZKL_ERR_INVALID;
    const int avail = ezkl_hip_context_count();
    if (avail < 1) return EZKL_ERR_NO_DEVICE;
    if (n_contexts > avail) return EZKL_ERR_INVALID;
    int world = 1;
    while (world * 2 <= n_contexts) world *= 2;              // the owner mode shards over a power of two
    auto grp = std::make_unique<ezkl_prover_group>();
    grp->world = world;
    grp->comm = std::make_unique<ThreadComm>(world);
    grp->cs.assign(world, nullptr);
    grp->g.assign(world, nullptr);
    grp->gl.assign(world, nullptr);
    grp->pk.assign(world, nullptr);
    grp->ranks.resize(world);
    int rc = EZKL_OK;
    for (int r = 0; r < world && !rc; r++) {
        grp->ranks[r] = ThreadRank{grp->comm.get(), r};
        rc = ezkl_prover_cs_parse(blob, len, &grp->cs[r]);
        if (rc || world == 1) continue;
        const uint32_t n = grp->cs[r]->cs->n, per = n / (uint32_t)world;
        if (per == 0) { rc = EZKL_ERR_INVALID; break; }
        rc = ezkl_prover_cs_set_shard(grp->cs[r], (uint32_t)r * per, (uint32_t)(r + 1) * per, tc_fold, &grp->ranks[r]);
        if (!rc) rc = ezkl_prover_cs_set_sweep_gather(grp->cs[r], tc_gather, &grp->ranks[r]);
        if (!rc) rc = ezkl_prover_cs_set_shard_exchange(grp->cs[r], tc_allgather_host, tc_exchange, &grp->ranks[r]);
        if (!rc) rc = ezkl_prover_cs_set_shard_full_bases(grp->cs[r], 1);
    }
    if (rc) {
        for (auto c : grp->cs) (void)ezkl_prover_cs_free(c);
        return rc;
    }
    *out = grp.release();
    return EZKL_OK;
}
int ezkl_prover_group_size(ezkl_group_t grp) { return grp ? grp->world : 0; }
int ezkl_prover_group_free(ezkl_group_t grp) {
    if (!grp) return EZKL_OK;
    (void)grp->run([&](int r) {                               // device objects are released on the context that made them
        if (grp->pk[r]) (void)ezkl_prover_pk_free(grp->pk[r]);
        if (grp->g[r]) (void)ezkl_hip_bases_free(grp->g[r]);
        if (grp->gl[r]) (void)ezkl_hip_bases_free(grp->gl[r]);
        return EZKL_OK;
    });
    for (auto c : grp->cs) (void)ezkl_prover_cs_free(c);
    delete grp;
    return EZKL_OK;
}
int ezkl_prover_group_load_srs(ezkl_group_t grp, const void* g_points, const void* g_lagrange_points, size_t n) {
    if (!grp || !g_points || !g_lagrange_points || n == 0) return EZKL_ERR_INVALID;
    return grp->run([&](int r) {
        if (grp->g[r]) { (void)ezkl_hip_bases_free(grp->g[r]); grp->g[r] = nullptr; }
        if (grp->gl[r]) { (void)ezkl_hip_bases_free(grp->gl[r]); grp->gl[r] = nullptr; }
        int rc = ezkl_hip_bases_upload(g_points, n, &grp->g[r]);
        if (!rc) rc = ezkl_hip_bases_upload(g_lagrange_points, n, &grp->gl[r]);
        if (!rc) rc = ezkl_hip_bases_prepare(grp->g[r]);
        if (!rc) rc = ezkl_hip_bases_prepare(grp->gl[r]);
        return rc;
    });
}
int ezkl_prover_group_keygen(ezkl_group_t grp, const void* const* fixed_values, const uint32_t* copies, size_t n_copies) {
    if (!grp) return EZKL_ERR_INVALID;
    return grp->run([&](int r) {
        if (!grp->g[r]) return (int)EZKL_ERR_INVALID;
        if (grp->pk[r]) { (void)ezkl_prover_pk_free(grp->pk[r]); grp->pk[r] = nullptr; }
        return ezkl_prover_keygen(grp->cs[r], grp->g[r], fixed_values, copies, n_copies, &grp->pk[r]);
    });
}
// load_pk for the group (/root/reference/src/pfsys/mod.rs:615-636: execute::prove READS pk.key, it does not run keygen): every context's
// thread maps the file and loads it through ezkl_prover_pk_read_file -- the n-row sections cross that context's PCIe link, the coefficient
// forms are recomputed on its device, and of the extended columns only the cosets that context sweeps are computed and kept (key bytes in
// HBM / world).  recommit != 0: the fixed / permutation commitments are recomputed under the group's SRS (a key file made with another SRS).
int ezkl_prover_group_pk_read_file(ezkl_group_t grp, const char* path, int recommit) {
    if (!grp || !path) return EZKL_ERR_INVALID;
    return grp->run([&](int r) {
        if (recommit && !grp->g[r]) return (int)EZKL_ERR_INVALID;
        if (grp->pk[r]) { (void)ezkl_prover_pk_free(grp->pk[r]); grp->pk[r] = nullptr; }
        int rc = ezkl_prover_pk_read_file(grp->cs[r], path, &grp->pk[r]);
        if (!rc && recommit) rc = ezkl_prover_pk_recommit(grp->pk[r], grp->g[r]);
        return rc;
    });
}
int ezkl_prover_group_pk(ezkl_group_t grp, int rank, ezkl_pk_t* out) {
    if (!grp || !out || rank < 0 || rank >= grp->world || !grp->pk[rank]) return EZKL_ERR_INVALID;
    *out = grp->pk[rank];
    return EZKL_OK;
}
int ezkl_prover_group_create_proof(ezkl_group_t grp, const void* const* advice, const void* const* instances, const uint32_t* instance_lens, uint64_t seed,
                                   void* proof_out, size_t cap, size_t* proof_len, double* timings, uint64_t* stats) {
    if (!grp || !advice || !proof_len) return EZKL_ERR_INVALID;
    // ONE source of randomness for every thread: the det-prove seed, or a 256-bit OS-entropy key drawn here
    Rng master(nullptr, nullptr, seed);
    std::vector<std::vector<uint8_t>> proofs(grp->world);
    std::vector<std::array<double, 12>> tm(grp->world);
    int rc = grp->run([&](int r) {
        if (!grp->pk[r] || !grp->g[r] || !grp->gl[r]) return (int)EZKL_ERR_INVALID;
        if (grp->pk[r]->pk->cs->n_instance && (!instances || !instance_lens)) return (int)EZKL_ERR_INVALID;
        return guarded([&] {
            Rng rng = master;
            proofs[r] = create_proof(*grp->pk[r]->pk, grp->g[r], grp->gl[r], advice, nullptr, nullptr, instances, instance_lens, rng, tm[r].data());
        });
    });
    if (rc) return rc;
    for (int r = 1; r < grp->world; r++)
        if (proofs[r] != proofs[0]) {
            g_last_error = "the contexts of the group produced different proofs";
            return EZKL_ERR_INVALID;
        }
    if (timings) {
        for (int i = 0; i < 12; i++) {
            double m = 0;
            for (int r = 0; r < grp->world; r++) m = std::max(m, tm[r][i]);
            timings[i] = m;
        }
    }
    if (stats)
        for (int r = 0; r < grp->world; r++) (void)ezkl_prover_cs_shard_stats(grp->cs[r], stats + 4 * r);
    *proof_len = proofs[0].size();
    if (proofs[0].size() > cap || !proof_out) {
        g_last_error = "proof buffer too small";
        return EZKL_ERR_NOMEM;
    }
    std::memcpy(proof_out, proofs[0].data(), proofs[0].size());
    return EZKL_OK;
}
}
