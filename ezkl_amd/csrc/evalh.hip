// evalh.hip -- row sweep of the quotient numerator: a straight-line field program evaluated on every row
// of the extended coset.  Replaces plonk::evaluation::GraphEvaluator::evaluate as driven by
// Evaluator::evaluate_h (halo2 fork pinned at /root/reference/Cargo.lock:2846-2848;
// the icicle build runs the same thing as a "gate_eval" program, Cargo.toml:99).
//
// Layout: columns are field-SoA in HBM (one contiguous 2^ext_k x 32 B array per column); a lane owns one
// row, so every column read is a unit-stride 32 B/lane stream and a rotation is just a different start
// offset ((r + rot * 2^(ext_k-k)) mod 2^ext_k).  The instruction stream, constants and challenges are
// wave-uniform (scalar loads); intermediates live in an HBM scratch laid out [slot][thread] so they
// stream as well.  Algorithmic bytes: 32*(C+1) per row.
#include "common.hpp"
#include "embedded_src.hpp"
#include <hip/hiprtc.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <functional>

namespace ezkl {

static constexpr uint32_t EV_NREG = 8;       // intermediates kept in VGPRs (wave-uniform register-file switch)

struct EvalArgs {
    const uint32_t* code;           // rewritten program: intermediate indices are slots (slot < EV_NREG: register)
    uint32_t n_instr;
    const fe_t* constants;
    const uint32_t* rot_off;        // rotation already scaled and reduced mod 2^ext_k
    const fe_t* const* columns;
    const fe_t* challenges;
    fe_t* interm;                   // spilled slots: [slot - EV_NREG][T]
    fe_t* out;
    uint32_t ne_mask;
    uint32_t T;
    uint32_t last_slot;
};

// The slot index is wave-uniform (it comes from the instruction stream), so these switches are scalar
// branches around 8 v_mov: ~30 cycles against the ~1000 of a Montgomery product.
#define EV_RD(dst, idx)                                                         \
    switch (idx) {                                                              \
    case 0: dst = r0; break; case 1: dst = r1; break; case 2: dst = r2; break; case 3: dst = r3; break; \
    case 4: dst = r4; break; case 5: dst = r5; break; case 6: dst = r6; break; default: dst = r7; break; }
#define EV_WR(idx, src)                                                         \
    switch (idx) {                                                              \
    case 0: r0 = src; break; case 1: r1 = src; break; case 2: r2 = src; break; case 3: r3 = src; break; \
    case 4: r4 = src; break; case 5: r5 = src; break; case 6: r6 = src; break; default: r7 = src; break; }

__global__ __launch_bounds__(256) void eval_program_kernel(EvalArgs a) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ne = a.ne_mask + 1;
    for (uint32_t r = tid; r < ne; r += a.T) {
        const fe_t prev = ld_fe(a.out + r);
        fe_t r0 = Fr::zero(), r1 = r0, r2 = r0, r3 = r0, r4 = r0, r5 = r0, r6 = r0, r7 = r0;
        for (uint32_t ii = 0; ii < a.n_instr; ii++) {
            const uint32_t* I = a.code + 8 * (size_t)ii;
            const uint32_t op = I[0], target = I[1];
            const bool unary = (op == EZKL_OP_SQUARE || op == EZKL_OP_DOUBLE || op == EZKL_OP_NEGATE || op == EZKL_OP_STORE);
            fe_t s[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (q == 1 && unary) break;
                const uint32_t kind = I[2 + 3 * q], idx = I[3 + 3 * q], rot = I[4 + 3 * q];
                switch (kind) {
                case EZKL_SRC_CONST: s[q] = ld_fe(a.constants + idx); break;
                case EZKL_SRC_INTERMEDIATE:
                    if (idx < EV_NREG) { EV_RD(s[q], idx) } else s[q] = ld_fe(a.interm + (size_t)(idx - EV_NREG) * a.T + tid);
                    break;
                case EZKL_SRC_COLUMN: s[q] = ld_fe(a.columns[idx] + ((r + a.rot_off[rot]) & a.ne_mask)); break;
                case EZKL_SRC_CHALLENGE: s[q] = ld_fe(a.challenges + idx); break;
                default: s[q] = prev; break;
                }
            }
            fe_t t;
            switch (op) {
            case EZKL_OP_ADD: t = Fr::add(s[0], s[1]); break;
            case EZKL_OP_SUB: t = Fr::sub(s[0], s[1]); break;
            case EZKL_OP_MUL: t = Fr::mul(s[0], s[1]); break;
            case EZKL_OP_SQUARE: t = Fr::mul(s[0], s[0]); break;
            case EZKL_OP_DOUBLE: t = Fr::dbl(s[0]); break;
            case EZKL_OP_NEGATE: t = Fr::neg(s[0]); break;
            case EZKL_OP_STORE: t = s[0]; break;
            default: {   // HORNER_STEP: target = target * s1 + s0
                fe_t cur;
                if (target < EV_NREG) { EV_RD(cur, target) } else cur = ld_fe(a.interm + (size_t)(target - EV_NREG) * a.T + tid);
                t = Fr::add(Fr::mul(cur, s[1]), s[0]);
            } break;
            }
            if (target < EV_NREG) { EV_WR(target, t) } else st_fe(a.interm + (size_t)(target - EV_NREG) * a.T + tid, t);
        }
        if (a.n_instr) {
            fe_t res;
            if (a.last_slot < EV_NREG) { EV_RD(res, a.last_slot) } else res = ld_fe(a.interm + (size_t)(a.last_slot - EV_NREG) * a.T + tid);
            st_fe(a.out + r, res);
        }
    }
}

// ---- instruction scheduling for short live ranges ----
// A host builds the numerator of h(X) the way halo2's GraphEvaluator does: every constraint term first, then ONE Horner chain over all
// of them (value = value * y + term).  In that order every term is live until the chain starts -- 99 terms x 8 VGPRs for a 48-gate /
// 12-lookup ezkl circuit: hiprtc's kernel came out with 256 VGPRs, 1984 bytes of scratch per lane and ONE wave per SIMD.  The order
// is not part of the program's meaning, so the library re-orders it: demand-driven from the consumers (each Horner step, then the
// final instruction), an instruction is emitted right before its first consumer.  Dependencies are tracked per intermediate INDEX
// (read-after-write, write-after-read, write-after-write), so any program a caller hands over keeps its value.
static std::vector<uint32_t> schedule_program(const ezkl_program_t* p) {
    const uint32_t n = p->n_instr, ni = p->n_intermediates;
    auto n_src = [](uint32_t op) { return (op == EZKL_OP_SQUARE || op == EZKL_OP_DOUBLE || op == EZKL_OP_NEGATE || op == EZKL_OP_STORE) ? 1 : 2; };
    std::vector<std::vector<uint32_t>> deps(n);
    std::vector<int64_t> last_writer(ni, -1);
    std::vector<std::vector<uint32_t>> readers(ni);          // readers of the current value of an intermediate
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t* I = p->code + 8 * (size_t)i;
        auto read = [&](uint32_t x) {
            if (last_writer[x] >= 0) deps[i].push_back((uint32_t)last_writer[x]);
            readers[x].push_back(i);
        };
        for (int q = 0; q < n_src(I[0]); q++)
            if (I[2 + 3 * q] == EZKL_SRC_INTERMEDIATE && I[3 + 3 * q] < ni) read(I[3 + 3 * q]);
        const uint32_t t = I[1];
        if (t >= ni) continue;                                // rejected later by the validation in eval_program
        if (I[0] == EZKL_OP_HORNER_STEP) read(t);             // updates its target in place
        for (uint32_t r : readers[t])
            if (r != i) deps[i].push_back(r);                 // write after read
        if (last_writer[t] >= 0) deps[i].push_back((uint32_t)last_writer[t]);   // write after write
        readers[t].clear();
        last_writer[t] = i;
    }
    std::vector<uint32_t> order;
    order.reserve(n);
    std::vector<uint8_t> state(n, 0);                         // 0 new, 1 on the stack, 2 emitted
    std::vector<std::pair<uint32_t, size_t>> stack;
    auto visit = [&](uint32_t root) {
        if (state[root]) return;
        stack.push_back({root, 0});
        state[root] = 1;
        while (!stack.empty()) {
            auto& top = stack.back();
            if (top.second < deps[top.first].size()) {
                const uint32_t d = deps[top.first][top.second++];
                if (!state[d]) { state[d] = 1; stack.push_back({d, 0}); }
            } else {
                order.push_back(top.first);
                state[top.first] = 2;
                stack.pop_back();
            }
        }
    };
    for (uint32_t i = 0; i + 1 < n; i++)
        if (p->code[8 * (size_t)i] == EZKL_OP_HORNER_STEP) visit(i);
    for (uint32_t i = 0; i + 1 < n; i++) visit(i);            // whatever no consumer asked for (dead code keeps its place before the result)
    if (n) visit(n - 1);                                      // the result is the last instruction's target: it stays last
    std::vector<uint32_t> code(8 * (size_t)n);
    for (uint32_t j = 0; j < n; j++) memcpy(&code[8 * (size_t)j], p->code + 8 * (size_t)order[j], 32);
    return code;
}

// Host-side register allocation: intermediates -> slots by linear scan over the straight-line program.
// A slot is released after the last read of its value; the EV_NREG lowest slots live in VGPRs, the rest spill
// to the HBM scratch.  Returns the rewritten code and the number of slots.
static uint32_t allocate_slots(const ezkl_program_t* p, std::vector<uint32_t>& code) {
    const uint32_t n = p->n_instr, ni = p->n_intermediates;
    code.assign(p->code, p->code + 8 * (size_t)n);
    auto n_src = [](uint32_t op) { return (op == EZKL_OP_SQUARE || op == EZKL_OP_DOUBLE || op == EZKL_OP_NEGATE || op == EZKL_OP_STORE) ? 1 : 2; };
    // pass 1: value versions.  A non-Horner write creates version i of its target; a Horner step updates
    // the current version in place.  last_use[ver] = index of the last instruction reading that version.
    std::vector<int64_t> cur_ver(ni, -1), last_use(n, -1);
    std::vector<int64_t> src_ver(2 * (size_t)n, -1), tgt_ver(n, -1);
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t* I = &code[8 * (size_t)i];
        for (int q = 0; q < n_src(I[0]); q++)
            if (I[2 + 3 * q] == EZKL_SRC_INTERMEDIATE) {
                int64_t v = cur_ver[I[3 + 3 * q]];
                src_ver[2 * (size_t)i + q] = v;            // -1: read of a never-written intermediate (reads zero)
                if (v >= 0) last_use[v] = i;
            }
        if (I[0] == EZKL_OP_HORNER_STEP && cur_ver[I[1]] >= 0) {
            tgt_ver[i] = cur_ver[I[1]];
            last_use[tgt_ver[i]] = i;
        } else {
            cur_ver[I[1]] = i;
            tgt_ver[i] = i;
        }
    }
    if (n) last_use[tgt_ver[n - 1]] = (int64_t)n;            // the result stays live to the end
    // pass 2: linear scan.  Sources are read into temporaries before the target is written, so a slot whose
    // value dies at instruction i can be reused by i's own target.
    std::vector<int64_t> slot_of(n, -1);
    std::vector<uint32_t> free_slots;
    uint32_t n_slots = 0;
    auto release = [&](int64_t ver) {
        if (ver >= 0 && slot_of[ver] >= 0) {
            free_slots.push_back((uint32_t)slot_of[ver]);
            slot_of[ver] = -2;
        }
    };
    auto acquire = [&]() -> uint32_t {
        if (free_slots.empty()) return n_slots++;
        auto it = std::min_element(free_slots.begin(), free_slots.end());   // lowest slot first: registers before spills
        uint32_t sl = *it;
        free_slots.erase(it);
        return sl;
    };
    for (uint32_t i = 0; i < n; i++) {
        uint32_t* I = &code[8 * (size_t)i];
        for (int q = 0; q < n_src(I[0]); q++)
            if (I[2 + 3 * q] == EZKL_SRC_INTERMEDIATE) {
                int64_t v = src_ver[2 * (size_t)i + q];
                if (v < 0) {                      // read of an intermediate no instruction has written: reject
                    return 0xffffffffu;
                } else {
                    I[3 + 3 * q] = (uint32_t)slot_of[v];
                }
            }
        const int64_t tv = tgt_ver[i];
        const bool in_place = (tv != (int64_t)i);
        for (int q = 0; q < n_src(I[0]); q++) {
            int64_t v = src_ver[2 * (size_t)i + q];
            if (v >= 0 && v != tv && last_use[v] == (int64_t)i) release(v);
        }
        if (!in_place) slot_of[tv] = (int64_t)acquire();
        I[1] = (uint32_t)slot_of[tv];
        if (last_use[tv] <= (int64_t)i && last_use[tv] != (int64_t)n) release(tv);    // dead after this instruction
    }
    return n_slots;
}

// ---- JIT path: the gate program becomes straight-line HIP source, compiled once per program with hiprtc ----
// An interpreter stalls on every instruction (decode -> dependent column load -> compute: ~2.4 us per
// instruction measured); in straight-line code the compiler hoists the column loads of a whole row ahead of
// the arithmetic and keeps every intermediate in VGPRs.  The kernel is cached by a hash of the program.
struct JitKernel {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    std::string key;          // the bytes the hash was taken of (code words + rotations): compared on a hit, a hash collision is a miss
    bool failed = false;      // negative cache: a program that did not compile is not compiled again on every sweep
};
// per-context state (Ctx::jit_state): a hipModule_t belongs to the device it was loaded on
struct JitState {
    std::multimap<uint64_t, JitKernel> jit;
    uint64_t compiled = 0, from_disk = 0, hits = 0;
};
static JitState& jit_state() {
    Ctx* c = ctx();
    if (!c->jit_state) c->jit_state = new JitState();
    return *static_cast<JitState*>(c->jit_state);
}
#define g_jit (jit_state().jit)
#define g_jit_compiled (jit_state().compiled)
#define g_jit_from_disk (jit_state().from_disk)
#define g_jit_hits (jit_state().hits)
void eval_jit_stats(uint64_t* compiled, uint64_t* from_disk, uint64_t* hits) { *compiled = g_jit_compiled; *from_disk = g_jit_from_disk; *hits = g_jit_hits; }

static uint64_t fnv1a(const void* data, size_t n, uint64_t h) {
    const uint8_t* p = (const uint8_t*)data;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
static int jit_knob(const char* name, int dflt) {
    const char* e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}
static std::string jit_source(const ezkl_program_t* p, const std::vector<uint32_t>& rot) {
    std::string s;
    s.reserve(256 + (size_t)p->n_instr * 96);
    s += "#include \"field.hpp\"\nusing namespace ezkl;\n";
    s += std::string("#define XCD_MAP ") + (jit_knob("EZKL_EVALH_XCD", 0) ? "1" : "0") + "\n";     // workgroups of one XCD walk adjacent rows
    const int waves = jit_knob("EZKL_EVALH_WAVES", 4), barrier = jit_knob("EZKL_EVALH_BARRIER", 1);
    s += "extern \"C\" __global__ __launch_bounds__(256) ";
    if (waves > 0) s += "__attribute__((amdgpu_waves_per_eu(" + std::to_string(waves) + "," + std::to_string(waves) + "))) ";
    s += "void evalh_jit(const fe_t* const* __restrict__ cols, const fe_t* __restrict__ consts,\n"
         "    const fe_t* __restrict__ chal, fe_t* __restrict__ out, uint32_t ne_mask, uint32_t T) {\n"
         "  const uint32_t tid = (XCD_MAP && gridDim.x % 8 == 0 ? (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x) * blockDim.x + threadIdx.x;\n"
         "  for (uint32_t r = tid; r <= ne_mask; r += T) {\n"
         "    const fe_t prev = ld_fe(out + r);\n";
    auto n_src = [](uint32_t op) { return (op == EZKL_OP_SQUARE || op == EZKL_OP_DOUBLE || op == EZKL_OP_NEGATE || op == EZKL_OP_STORE) ? 1 : 2; };
    std::vector<int64_t> cur_ver(p->n_intermediates, -1);
    auto src = [&](const uint32_t* I, int q) -> std::string {
        const uint32_t kind = I[2 + 3 * q], idx = I[3 + 3 * q], ro = I[4 + 3 * q];
        char b[96];
        switch (kind) {
        case EZKL_SRC_CONST: snprintf(b, sizeof b, "ld_fe(consts + %u)", idx); break;
        case EZKL_SRC_INTERMEDIATE: snprintf(b, sizeof b, "v%lld", (long long)cur_ver[idx]); break;
        case EZKL_SRC_COLUMN: snprintf(b, sizeof b, "ld_fe(cols[%u] + ((r + %uu) & ne_mask))", idx, rot[ro]); break;
        case EZKL_SRC_CHALLENGE: snprintf(b, sizeof b, "ld_fe(chal + %u)", idx); break;
        default: snprintf(b, sizeof b, "prev"); break;
        }
        return b;
    };
    long long last = -1;
    for (uint32_t i = 0; i < p->n_instr; i++) {
        const uint32_t* I = p->code + 8 * (size_t)i;
        std::string a = src(I, 0), b = n_src(I[0]) == 2 ? src(I, 1) : std::string();
        char lhs[48];
        if (I[0] == EZKL_OP_HORNER_STEP && cur_ver[I[1]] >= 0) {
            snprintf(lhs, sizeof lhs, "    v%lld = ", (long long)cur_ver[I[1]]);
            s += lhs;
            s += "Fr::add(Fr::mul(v" + std::to_string(cur_ver[I[1]]) + ", " + b + "), " + a + ");\n";
            if (barrier == 1) s += "    asm volatile(\"\" ::: \"memory\");\n";
            else if (barrier == 2) s += "    __builtin_amdgcn_sched_barrier(0);\n";
            last = cur_ver[I[1]];
            continue;
        }
        cur_ver[I[1]] = i;
        last = i;
        snprintf(lhs, sizeof lhs, "    fe_t v%u = ", i);
        s += lhs;
        switch (I[0]) {
        case EZKL_OP_ADD: s += "Fr::add(" + a + ", " + b + ");\n"; break;
        case EZKL_OP_SUB: s += "Fr::sub(" + a + ", " + b + ");\n"; break;
        case EZKL_OP_MUL: s += "Fr::mul(" + a + ", " + b + ");\n"; break;
        case EZKL_OP_SQUARE: s += "Fr::sqr(" + a + ");\n"; break;
        case EZKL_OP_DOUBLE: s += "Fr::dbl(" + a + ");\n"; break;
        case EZKL_OP_NEGATE: s += "Fr::neg(" + a + ");\n"; break;
        case EZKL_OP_STORE: s += a + ";\n"; break;
        default: s += "Fr::add(Fr::mul(Fr::zero(), " + b + "), " + a + ");\n"; break;   // Horner on an unwritten target
        }
    }
    s += "    st_fe(out + r, v" + std::to_string(last) + ");\n  }\n}\n";
    return s;
}
// ---- the same program in radix 2^29 (field29.hpp): the product costs 186 issue slots instead of 261 ----
// Values live in the lazily reduced Montgomery form R' = 2^261 of the MSM kernels.  A column element x*R (canonical, < p) becomes a
// representative of x*R' by a 5-bit shift while it is unpacked (32 * x*R < 32 p: a valid lazy value, no multiplication), every
// intermediate carries, at CODE-GENERATION time, a bound alpha (value < alpha * p) and a limb looseness L (limbs < L * 2^29), and the
// generator inserts a carry propagation or a multiplication by one only where the rules of field29.hpp need it:
//   add: limb-wise, alpha and L add up (kept <= 160 and <= 6);  sub: a + (K p - b) with the smallest K in {2 .. 128} above b's bound,
//   b normalized;  mul: L_a * L_b <= 6, alpha_a * alpha_b <= 5000, result < (alpha_a alpha_b / 169 + 1) p, normalized.
// The result is multiplied by 2^256 (R' -> R), reduced to [0, p) and packed: the bytes written are those of the radix-2^32 kernel.
struct V29 {
    std::string expr;      // a variable name, or a load expression
    int alpha = 32, L = 1;
    long long var = -1;    // version id when `expr` is a variable (its state lives in the table), -1 for a load
};
static std::string jit_source_r29(const ezkl_program_t* p, const std::vector<uint32_t>& rot) {
    std::string s;
    s.reserve(1024 + (size_t)p->n_instr * 160);
    const int waves = jit_knob("EZKL_EVALH_WAVES", 4), barrier = jit_knob("EZKL_EVALH_BARRIER", 1);
    const std::string MUL = jit_knob("EZKL_EVALH_R29", 2) == 2 ? "Fr29::mul_cold(" : "Fr29::mul(";      // 2: the product as a call (small code)
    // the compiler barrier that makes every term reload its columns (no scratch, 4 waves per SIMD) after every N-th Horner step: N > 1 lets
    // neighbouring terms share loaded values at the price of registers
    const int barrier_every = jit_knob("EZKL_EVALH_BARRIER_EVERY", 4) > 0 ? jit_knob("EZKL_EVALH_BARRIER_EVERY", 4) : 1;     // 4: k = 20 MLP sweep 2.54 -> 2.41 ms per coset (profiles/r04p_evalh_ab.log)
    int horner_steps = 0;
    s += std::string("#define XCD_MAP ") + (jit_knob("EZKL_EVALH_XCD", 0) ? "1" : "0") + "\n";
    s += "#include \"field29.hpp\"\nusing namespace ezkl;\n"
         "__device__ __forceinline__ f29_t ld29(const fe_t* q) {\n"
         "  const fe_t w = ld_fe(q);\n  f29_t r;\n  r.v[0] = (w.v[0] << 5) & M29;\n"
         "#pragma unroll\n  for (int i = 1; i < 8; i++) {\n"
         "    const int bit = 29 * i - 5, word = bit >> 5, sh = bit & 31;\n"
         "    const uint32_t lo = w.v[word], hi = word + 1 < 8 ? w.v[word + 1] : 0u;\n"
         "    r.v[i] = (sh ? __builtin_amdgcn_alignbit(hi, lo, sh) : lo) & M29;\n  }\n"
         "  r.v[8] = w.v[7] >> 3;\n  return r;\n}\n";
    s += "extern \"C\" __global__ __launch_bounds__(256) ";
    if (waves > 0) s += "__attribute__((amdgpu_waves_per_eu(" + std::to_string(waves) + "," + std::to_string(waves) + "))) ";
    s += "void evalh_jit(const fe_t* const* __restrict__ cols, const fe_t* __restrict__ consts,\n"
         "    const fe_t* __restrict__ chal, fe_t* __restrict__ out, uint32_t ne_mask, uint32_t T) {\n"
         "  const uint32_t tid = (XCD_MAP && gridDim.x % 8 == 0 ? (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x) * blockDim.x + threadIdx.x;\n"
         "  const f29_t c_one = Fr29::one();\n"
         "  const f29_t c_r256 = Fr29::unpack(Fr::one());\n"        // 2^256 mod p, normalized: R' -> R
         "  for (uint32_t r = tid; r <= ne_mask; r += T) {\n";
    auto n_src = [](uint32_t op) { return (op == EZKL_OP_SQUARE || op == EZKL_OP_DOUBLE || op == EZKL_OP_NEGATE || op == EZKL_OP_STORE) ? 1 : 2; };
    std::vector<int64_t> cur_ver(p->n_intermediates, -1);
    std::vector<int> A(p->n_instr, 0), Ls(p->n_instr, 0);                 // state of version i (defined by instruction i)
    auto operand = [&](const uint32_t* I, int q) -> V29 {
        const uint32_t kind = I[2 + 3 * q], idx = I[3 + 3 * q], ro = I[4 + 3 * q];
        char b[112];
        V29 v;
        switch (kind) {
        case EZKL_SRC_CONST: snprintf(b, sizeof b, "ld29(consts + %u)", idx); break;
        case EZKL_SRC_INTERMEDIATE: {
            const long long ver = cur_ver[idx];
            snprintf(b, sizeof b, "v%lld", ver);
            v.var = ver;
            v.alpha = A[ver];
            v.L = Ls[ver];
            break;
        }
        case EZKL_SRC_COLUMN: snprintf(b, sizeof b, "ld29(cols[%u] + ((r + %uu) & ne_mask))", idx, rot[ro]); break;
        case EZKL_SRC_CHALLENGE: snprintf(b, sizeof b, "ld29(chal + %u)", idx); break;
        default: snprintf(b, sizeof b, "ld29(out + r)"); break;
        }
        v.expr = b;
        return v;
    };
    auto normalize = [&](V29& v) {                     // carry propagation: limbs back below 2^29
        if (v.L == 1) return;
        s += "    " + v.expr + " = Fr29::normalize(" + v.expr + ");\n";
        v.L = 1;
        Ls[v.var] = 1;
    };
    auto reduce = [&](V29& v) {                        // a multiplication by one: the bound falls to alpha / 169 + 2
        if (v.var < 0) return;                         // loads are < 32 p by construction
        if (v.L > 6) normalize(v);
        s += "    " + v.expr + " = " + MUL + v.expr + ", c_one);\n";
        v.alpha = v.alpha / 169 + 2;
        v.L = 1;
        A[v.var] = v.alpha;
        Ls[v.var] = 1;
    };
    auto prep_mul = [&](V29& a, V29& b) {
        while (a.L * b.L > 6) normalize(a.L >= b.L && a.var >= 0 ? a : b);
        while ((long long)a.alpha * b.alpha > 5000) reduce(a.alpha >= b.alpha && a.var >= 0 ? a : b);
        return ((long long)a.alpha * b.alpha + 168) / 169 + 1;
    };
    auto prep_add = [&](V29& a, V29& b) {
        while (a.L + b.L > 6) normalize(a.L >= b.L && a.var >= 0 ? a : b);
        while (a.alpha + b.alpha > 160) reduce(a.alpha >= b.alpha && a.var >= 0 ? a : b);
    };
    auto sub_k = [&](V29& a, V29& b, int& ki) {        // a - b + K p
        normalize(b);
        if (b.alpha > 127) reduce(b);
        int K = 2;
        ki = 0;
        while (K - 1 < b.alpha) { K <<= 1; ki++; }
        while (a.L + 2 > 6) normalize(a);
        while (a.alpha + K > 160 && a.var >= 0 && a.alpha > 3) reduce(a);
        return K;
    };
    long long last = -1;
    for (uint32_t i = 0; i < p->n_instr; i++) {
        const uint32_t* I = p->code + 8 * (size_t)i;
        V29 a = operand(I, 0), b = n_src(I[0]) == 2 ? operand(I, 1) : V29();
        if (I[0] == EZKL_OP_HORNER_STEP && cur_ver[I[1]] >= 0) {
            const long long tv = cur_ver[I[1]];
            V29 t;
            t.expr = "v" + std::to_string(tv);
            t.var = tv; t.alpha = A[tv]; t.L = Ls[tv];
            const long long am = prep_mul(t, b);
            // the sum acc * factor + term: the product is normalized (L = 1) and below am * p
            while (1 + a.L > 6) normalize(a);
            while (am + a.alpha > 160 && a.var >= 0) reduce(a);
            s += "    " + t.expr + " = Fr29::add(" + MUL + t.expr + ", " + b.expr + "), " + a.expr + ");\n";
            if (++horner_steps % barrier_every == 0) {
                if (barrier == 1) s += "    asm volatile(\"\" ::: \"memory\");\n";
                else if (barrier == 2) s += "    __builtin_amdgcn_sched_barrier(0);\n";
            }
            A[tv] = (int)(am + a.alpha);
            Ls[tv] = 1 + a.L;
            last = tv;
            continue;
        }
        cur_ver[I[1]] = i;
        last = i;
        const std::string lhs = "    f29_t v" + std::to_string(i) + " = ";
        switch (I[0]) {
        case EZKL_OP_ADD:
            prep_add(a, b);
            s += lhs + "Fr29::add(" + a.expr + ", " + b.expr + ");\n";
            A[i] = a.alpha + b.alpha; Ls[i] = a.L + b.L;
            break;
        case EZKL_OP_DOUBLE:
            while (2 * a.L > 6) normalize(a);
            while (2 * a.alpha > 160) reduce(a);
            s += lhs + "Fr29::add(" + a.expr + ", " + a.expr + ");\n";
            A[i] = 2 * a.alpha; Ls[i] = 2 * a.L;
            break;
        case EZKL_OP_SUB: {
            int ki = 0;
            const int K = sub_k(a, b, ki);
            s += lhs + "Fr29::sub<" + std::to_string(ki) + ">(" + a.expr + ", " + b.expr + ");\n";
            A[i] = a.alpha + K; Ls[i] = a.L + 2;
            break;
        }
        case EZKL_OP_NEGATE: {
            normalize(a);
            if (a.alpha > 127) reduce(a);
            int K = 2, ki = 0;
            while (K - 1 < a.alpha) { K <<= 1; ki++; }
            s += lhs + "Fr29::neg<" + std::to_string(ki) + ">(" + a.expr + ");\n";
            A[i] = K; Ls[i] = 2;
            break;
        }
        case EZKL_OP_MUL: {
            const long long am = prep_mul(a, b);
            s += lhs + MUL + a.expr + ", " + b.expr + ");\n";
            A[i] = (int)am; Ls[i] = 1;
            break;
        }
        case EZKL_OP_SQUARE: {
            while (a.L > 2) normalize(a);
            while ((long long)a.alpha * a.alpha > 5000) reduce(a);
            s += lhs + MUL + a.expr + ", " + a.expr + ");\n";
            A[i] = (int)(((long long)a.alpha * a.alpha + 168) / 169 + 1); Ls[i] = 1;
            break;
        }
        case EZKL_OP_STORE:
            s += lhs + a.expr + ";\n";
            A[i] = a.alpha; Ls[i] = a.L;
            break;
        default:                                       // Horner on an unwritten target: 0 * factor + term
            s += lhs + a.expr + ";\n";
            A[i] = a.alpha; Ls[i] = a.L;
            break;
        }
    }
    // R' -> R and canonical form: v * 2^256 / 2^261 < (alpha / 169 + 1) p <= 2 p, one conditional subtraction, pack
    {
        V29 v;
        v.expr = "v" + std::to_string(last);
        v.var = last; v.alpha = A[last]; v.L = Ls[last];
        while (v.L > 6) normalize(v);
        while (v.alpha > 169) reduce(v);
        s += "    st_fe(out + r, Fr29::pack(Fr29::cond_sub<0>(" + MUL + v.expr + ", c_r256))));\n  }\n}\n";
    }
    return s;
}
// Compiled code objects are also kept on disk, keyed by a hash of (program, rotations, architecture, library build): the gate program
// of a circuit is the same in every `ezkl prove` process, and a one-shot prover (src/execute.rs:1575-1627) should not pay hiprtc
// (0.2-1 s per program) each time.  Directory: $EZKL_HIP_CACHE_DIR, else $XDG_CACHE_HOME/ezkl_hip, else ~/.cache/ezkl_hip;
// EZKL_HIP_CACHE_DIR=off disables it.  Files are written to a temporary name and renamed (concurrent provers).
static std::string jit_cache_dir() {
    const char* e = getenv("EZKL_HIP_CACHE_DIR");
    if (e && (!strcmp(e, "off") || !*e)) return "";
    std::string d;
    if (e) d = e;
    else if (const char* x = getenv("XDG_CACHE_HOME")) d = std::string(x) + "/ezkl_hip";
    else if (const char* h = getenv("HOME")) d = std::string(h) + "/.cache/ezkl_hip";
    else return "";
    std::string cur;
    for (size_t i = 0; i <= d.size(); i++) {       // mkdir -p
        if (i == d.size() || (d[i] == '/' && i)) {
            cur = d.substr(0, i);
            if (!cur.empty()) (void)mkdir(cur.c_str(), 0700);
        }
    }
    return d;
}
static std::string jit_arch(Ctx* c) {
    hipDeviceProp_t prop;
    if (c && hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.gcnArchName[0]) {
        std::string a = prop.gcnArchName;          // "gfx950:sramecc+:xnack-" -> "gfx950"
        return a.substr(0, a.find(':'));
    }
    return "gfx950";
}
// What the generated code depends on besides the program: the headers hiprtc compiles it against (embedded at build time) and the code
// generators in this file.  Their digest is part of the cache identity, so a code object written by another build of the library -- other
// field arithmetic, other generator -- is never loaded for the same program (the version string alone did not change).
static constexpr uint32_t JIT_CODEGEN_REVISION = 3;     // bump when jit_source / jit_source_r29 / schedule_program change what they emit
static uint64_t jit_build_digest() {
    static const uint64_t d = [] {
        uint64_t h = 1469598103934665603ull ^ JIT_CODEGEN_REVISION;
        for (const char* src : {k_src_field, k_src_constants, k_src_montmul, k_src_field29, k_src_montmul29}) h = fnv1a(src, strlen(src), h * 31 + 7);
        return fnv1a(ezkl_hip_version(), strlen(ezkl_hip_version()), h);
    }();
    return d;
}
static int jit_get(Ctx* c, const ezkl_program_t* p, const std::vector<uint32_t>& rot, hipFunction_t* fn) {
    std::string key((const char*)p->code, (size_t)p->n_instr * 32);
    key.append((const char*)rot.data(), rot.size() * 4);
    const uint64_t build = jit_build_digest();
    key.append((const char*)&build, sizeof build);
    const int knobs[5] = {jit_knob("EZKL_EVALH_WAVES", 4), jit_knob("EZKL_EVALH_BARRIER", 1), jit_knob("EZKL_EVALH_R29", 2), jit_knob("EZKL_EVALH_XCD", 0),
                          jit_knob("EZKL_EVALH_BARRIER_EVERY", 4)};      // code-generation options are part of the identity
    key.append((const char*)knobs, sizeof knobs);
    const uint64_t h = fnv1a(key.data(), key.size(), 1469598103934665603ull);
    if (getenv("EZKL_HIP_JIT_DEBUG")) fprintf(stderr, "[ezkl_hip] sweep kernel %016llx: %u instructions, %u columns, ext_k %u\n", (unsigned long long)h, p->n_instr, p->n_columns, p->ext_k);
    auto range = g_jit.equal_range(h);
    for (auto it = range.first; it != range.second; ++it) {
        if (it->second.key != key) continue;       // a 64-bit collision: not this program
        if (it->second.failed) return EZKL_ERR_HIP;
        *fn = it->second.fn;
        g_jit_hits++;
        return EZKL_OK;
    }
    JitKernel k;
    k.key = key;
    const std::string arch = jit_arch(c), dir = jit_cache_dir();
    char name[160];
    // second, independent hash in the file name + the key length: a stale or colliding file is caught by the embedded key check below
    snprintf(name, sizeof name, "/evalh_%s_%016llx_%016llx_%zu.co", arch.c_str(), (unsigned long long)h,
             (unsigned long long)fnv1a(key.data(), key.size(), 0x9e3779b97f4a7c15ull ^ build), key.size());
    std::vector<char> bin;
    bool from_disk = false;
    if (!dir.empty()) {
        if (FILE* f = fopen((dir + name).c_str(), "rb")) {
            fseek(f, 0, SEEK_END);
            long sz = ftell(f);
            fseek(f, 0, SEEK_SET);
            if (sz > (long)key.size() + 8) {
                std::vector<char> all((size_t)sz);
                if (fread(all.data(), 1, (size_t)sz, f) == (size_t)sz && !memcmp(all.data(), key.data(), key.size())) {
                    bin.assign(all.begin() + key.size(), all.end());
                    from_disk = true;
                }
            }
            fclose(f);
        }
    }
    if (!from_disk) {
        std::string src = jit_knob("EZKL_EVALH_R29", 2) ? jit_source_r29(p, rot) : jit_source(p, rot);
        const char* hn[5] = {"field.hpp", "bn254_constants.h", "montmul_gen.hpp", "field29.hpp", "montmul29_gen.hpp"};
        const char* hs[5] = {k_src_field, k_src_constants, k_src_montmul, k_src_field29, k_src_montmul29};
        hiprtcProgram prog;
        if (hiprtcCreateProgram(&prog, src.c_str(), "evalh_jit.hip", 5, hs, hn) != HIPRTC_SUCCESS) return EZKL_ERR_HIP;
        const std::string archopt = "--offload-arch=" + arch;
        const char* opts[] = {archopt.c_str(), "-O3", "-std=c++17"};
        hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
        if (r != HIPRTC_SUCCESS) {
            size_t ls = 0;
            hiprtcGetProgramLogSize(prog, &ls);
            std::string log(ls, 0);
            if (ls) hiprtcGetProgramLog(prog, &log[0]);
            fprintf(stderr, "[ezkl_hip] eval_h JIT compile failed (%d):\n%.2000s\n", (int)r, log.c_str());
            hiprtcDestroyProgram(&prog);
            k.failed = true;
            g_jit.emplace(h, k);
            return EZKL_ERR_HIP;
        }
        size_t cs = 0;
        hiprtcGetCodeSize(prog, &cs);
        bin.resize(cs);
        hiprtcGetCode(prog, bin.data());
        hiprtcDestroyProgram(&prog);
        if (!dir.empty()) {
            char tmp[64];
            snprintf(tmp, sizeof tmp, ".tmp.%d.%llx", (int)getpid(), (unsigned long long)h);
            const std::string tpath = dir + name + tmp;
            if (FILE* f = fopen(tpath.c_str(), "wb")) {
                const bool ok = fwrite(key.data(), 1, key.size(), f) == key.size() && fwrite(bin.data(), 1, bin.size(), f) == bin.size();
                fclose(f);
                if (!ok || rename(tpath.c_str(), (dir + name).c_str()) != 0) (void)remove(tpath.c_str());
            }
        }
    }
    if (hipModuleLoadData(&k.mod, bin.data()) != hipSuccess || hipModuleGetFunction(&k.fn, k.mod, "evalh_jit") != hipSuccess) {
        (void)hipGetLastError();
        if (from_disk) (void)remove((dir + name).c_str());     // a damaged cache entry: drop it, the next call compiles
        k.failed = !from_disk;
        if (k.failed) g_jit.emplace(h, k);
        return EZKL_ERR_HIP;
    }
    g_jit.emplace(h, k);
    *fn = k.fn;
    if (from_disk) g_jit_from_disk++; else g_jit_compiled++;
    return EZKL_OK;
}
// every entry point validates the WHOLE program -- opcodes, targets and every source operand -- before anything indexes with it
// (schedule_program / allocate_slots index by operand, the generators index constants and columns)
static int validate_program(const ezkl_program_t* p) {
    if (p->ext_k > 28 || p->k > p->ext_k) return EZKL_ERR_INVALID;
    if (p->n_instr && !p->code) return EZKL_ERR_INVALID;
    for (uint32_t i = 0; i < p->n_instr; i++) {
        const uint32_t* I = p->code + 8 * (size_t)i;
        if (I[0] > EZKL_OP_HORNER_STEP || I[1] >= p->n_intermediates) return EZKL_ERR_INVALID;
        const bool unary = (I[0] == EZKL_OP_SQUARE || I[0] == EZKL_OP_DOUBLE || I[0] == EZKL_OP_NEGATE || I[0] == EZKL_OP_STORE);
        for (int q = 0; q < (unary ? 1 : 2); q++) {
            uint32_t kind = I[2 + 3 * q], idx = I[3 + 3 * q], rot = I[4 + 3 * q];
            if (kind > EZKL_SRC_PREVIOUS) return EZKL_ERR_INVALID;
            if (kind == EZKL_SRC_CONST && idx >= p->n_constants) return EZKL_ERR_INVALID;
            if (kind == EZKL_SRC_INTERMEDIATE && idx >= p->n_intermediates) return EZKL_ERR_INVALID;
            if (kind == EZKL_SRC_COLUMN && (idx >= p->n_columns || rot >= p->n_rotations)) return EZKL_ERR_INVALID;
            if (kind == EZKL_SRC_CHALLENGE && idx >= p->n_challenges) return EZKL_ERR_INVALID;
        }
    }
    return EZKL_OK;
}
// offline self-check used by build(): does the JIT source for a program compile for gfx950? (no GPU needed)
int eval_jit_compile_only(const ezkl_program_t* p0) {
    if (int rc = validate_program(p0)) return rc;
    ezkl_program_t scheduled = *p0;
    const std::vector<uint32_t> sched_code = getenv("EZKL_EVALH_NO_SCHEDULE") ? std::vector<uint32_t>(p0->code, p0->code + 8 * (size_t)p0->n_instr) : schedule_program(p0);
    scheduled.code = sched_code.data();
    const ezkl_program_t* p = &scheduled;
    {
        std::vector<uint32_t> tmp;
        if (allocate_slots(p, tmp) == 0xffffffffu) return EZKL_ERR_INVALID;
    }
    std::vector<uint32_t> rot(p->n_rotations ? p->n_rotations : 1, 0);
    std::string src = jit_knob("EZKL_EVALH_R29", 2) ? jit_source_r29(p, rot) : jit_source(p, rot);
    if (const char* dump = getenv("EZKL_HIP_JIT_DUMP")) {     // developer aid: the generated source, to look at its register use offline
        if (FILE* f = fopen(dump, "w")) { fwrite(src.data(), 1, src.size(), f); fclose(f); }
    }
    const char* hn[5] = {"field.hpp", "bn254_constants.h", "montmul_gen.hpp", "field29.hpp", "montmul29_gen.hpp"};
    const char* hs[5] = {k_src_field, k_src_constants, k_src_montmul, k_src_field29, k_src_montmul29};
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "evalh_jit.hip", 5, hs, hn) != HIPRTC_SUCCESS) return EZKL_ERR_HIP;
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
    hiprtcDestroyProgram(&prog);
    return r == HIPRTC_SUCCESS ? EZKL_OK : EZKL_ERR_HIP;
}

// host-only: the order the library will execute a program in (schedule_program), for callers and tests that want to look at it
int eval_schedule_only(const ezkl_program_t* p, uint32_t* out_code) {
    if (int rc = validate_program(p)) return rc;
    const std::vector<uint32_t> code = schedule_program(p);
    memcpy(out_code, code.data(), code.size() * 4);
    return EZKL_OK;
}

// Compile (or load from the on-disk cache) the kernel of a program WITHOUT running it: a key generator calls this for the circuit's
// quotient program, so that the first `prove` of a new circuit does not wait for hiprtc (9 s for a 786-instruction ezkl program).
int eval_prepare(Ctx* c, const ezkl_program_t* p0) {
    if (p0->n_instr == 0) return EZKL_ERR_INVALID;
    if (int rc = validate_program(p0)) return rc;
    ezkl_program_t scheduled = *p0;
    const std::vector<uint32_t> sched_code = getenv("EZKL_EVALH_NO_SCHEDULE") ? std::vector<uint32_t>(p0->code, p0->code + 8 * (size_t)p0->n_instr) : schedule_program(p0);
    scheduled.code = sched_code.data();
    {
        std::vector<uint32_t> tmp;                      // rejects reads of intermediates no instruction has written (the generators index by version)
        if (allocate_slots(&scheduled, tmp) == 0xffffffffu) return EZKL_ERR_INVALID;
    }
    const size_t ne = (size_t)1 << p0->ext_k;
    std::vector<uint32_t> rot(p0->n_rotations ? p0->n_rotations : 1, 0);
    const int64_t scale = (int64_t)1 << (p0->ext_k - p0->k);
    for (uint32_t i = 0; i < p0->n_rotations; i++) {
        int64_t v = ((int64_t)p0->rotations[i] * scale) % (int64_t)ne;
        if (v < 0) v += (int64_t)ne;
        rot[i] = (uint32_t)v;
    }
    hipFunction_t fn = nullptr;
    return jit_get(c, &scheduled, rot, &fn);
}

// `ordered`: the call is stream-ordered (a caller stream, or the library stream in asynchronous mode) and returns once queued
int eval_program(Ctx* c, hipStream_t st, const ezkl_program_t* p, fe_t* out, bool ordered) {
    if (int vrc = validate_program(p)) return vrc;          // before it touches the device
    if (p->n_instr == 0) return EZKL_OK;
    const size_t ne = (size_t)1 << p->ext_k;
    ezkl_program_t scheduled = *p;
    const std::vector<uint32_t> sched_code = getenv("EZKL_EVALH_NO_SCHEDULE") ? std::vector<uint32_t>(p->code, p->code + 8 * (size_t)p->n_instr) : schedule_program(p);
    scheduled.code = sched_code.data();
    p = &scheduled;                                           // from here on: the same program in an order with short live ranges
    std::vector<uint32_t> code;
    const uint32_t n_slots = allocate_slots(p, code);
    if (n_slots == 0xffffffffu) return EZKL_ERR_INVALID;
    const uint32_t n_spill = n_slots > EV_NREG ? n_slots - EV_NREG : 0;
    // resident lanes: one workgroup of 4 waves per SIMD-wave the kernel is compiled for (EZKL_EVALH_WAVES), so that the grid can
    // actually fill the occupancy the code generator asked for; rows beyond that are walked by the kernel's loop
    size_t T = (size_t)c->num_cus * 256 * (size_t)jit_knob("EZKL_EVALH_TMUL", 16);
    if (T > ne) T = ne;
    std::vector<uint32_t> rot(p->n_rotations ? p->n_rotations : 1, 0);
    const int64_t scale = (int64_t)1 << (p->ext_k - p->k);
    for (uint32_t i = 0; i < p->n_rotations; i++) {
        int64_t v = ((int64_t)p->rotations[i] * scale) % (int64_t)ne;
        if (v < 0) v += (int64_t)ne;
        rot[i] = (uint32_t)v;
    }
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o_code = 0;
    size_t o_const = o_code + al((size_t)p->n_instr * 32);
    size_t o_rot = o_const + al((size_t)(p->n_constants ? p->n_constants : 1) * 32);
    size_t o_cols = o_rot + al(rot.size() * 4);
    size_t o_chal = o_cols + al((size_t)(p->n_columns ? p->n_columns : 1) * 8);
    size_t o_int = o_chal + al((size_t)(p->n_challenges ? p->n_challenges : 1) * 32);
    size_t total = o_int + al((size_t)(n_spill ? n_spill : 1) * T * 32);
    uint8_t* S = nullptr;
    int rc = arena_reserve(scratch_arena(c, st), total, st, (void**)&S);
    if (rc) return rc;
    // The caller's arrays are borrowed only for the call: they are packed into a pinned block that stays valid until the copy engine
    // has read it, so a host can queue many sweeps / helper programs without waiting for any of them (the lookup and permutation
    // phases of a proof run ~20 small programs back to back)
    void* stg = nullptr;
    uint8_t* H = staging_acquire(c, o_int, &stg);
    if (H) {                                                   // one copy from a pinned block the library owns
        memcpy(H + o_code, code.data(), (size_t)p->n_instr * 32);
        if (p->n_constants) memcpy(H + o_const, p->constants, (size_t)p->n_constants * 32);
        memcpy(H + o_rot, rot.data(), rot.size() * 4);
        if (p->n_columns) memcpy(H + o_cols, p->columns, (size_t)p->n_columns * 8);
        if (p->n_challenges) memcpy(H + o_chal, p->challenges, (size_t)p->n_challenges * 32);
        EZ_HIP(hipMemcpyAsync(S, H, o_int, hipMemcpyHostToDevice, st));
        if ((rc = staging_release(stg, st))) return rc;
    } else {
        ordered = false;                                       // the arrays are read where they lie: synchronise below
        EZ_HIP(hipMemcpyAsync(S + o_code, code.data(), (size_t)p->n_instr * 32, hipMemcpyHostToDevice, st));
        if (p->n_constants) EZ_HIP(hipMemcpyAsync(S + o_const, p->constants, (size_t)p->n_constants * 32, hipMemcpyHostToDevice, st));
        EZ_HIP(hipMemcpyAsync(S + o_rot, rot.data(), rot.size() * 4, hipMemcpyHostToDevice, st));
        if (p->n_columns) EZ_HIP(hipMemcpyAsync(S + o_cols, p->columns, (size_t)p->n_columns * 8, hipMemcpyHostToDevice, st));
        if (p->n_challenges) EZ_HIP(hipMemcpyAsync(S + o_chal, p->challenges, (size_t)p->n_challenges * 32, hipMemcpyHostToDevice, st));
    }
    EvalArgs a;
    a.code = (const uint32_t*)(S + o_code);
    a.n_instr = p->n_instr;
    a.constants = (const fe_t*)(S + o_const);
    a.rot_off = (const uint32_t*)(S + o_rot);
    a.columns = (const fe_t* const*)(S + o_cols);
    a.challenges = (const fe_t*)(S + o_chal);
    a.interm = (fe_t*)(S + o_int);
    a.out = out;
    a.ne_mask = (uint32_t)(ne - 1);
    a.T = (uint32_t)T;
    a.last_slot = code[8 * (size_t)(p->n_instr - 1) + 1];
    hipEvent_t e0, e1;
    if ((rc = ev_pair(c, "eval_h", &e0, &e1))) return rc;
    const char* mode = getenv("EZKL_EVALH_MODE");                 // "interp" forces the interpreter
    hipFunction_t jfn = nullptr;
    const bool use_jit = !(mode && !strcmp(mode, "interp")) && jit_get(c, p, rot, &jfn) == EZKL_OK;
    EZ_HIP(hipEventRecord(e0, st));
    if (use_jit) {
        const fe_t* const* d_cols = a.columns;
        const fe_t* d_consts = a.constants;
        const fe_t* d_chal = a.challenges;
        fe_t* d_out = out;
        uint32_t ne_mask = a.ne_mask, Tj = a.T;
        void* args[] = {&d_cols, &d_consts, &d_chal, &d_out, &ne_mask, &Tj};
        EZ_HIP(hipModuleLaunchKernel(jfn, cdiv(T, 256), 1, 1, 256, 1, 1, 0, st, args, nullptr));
    } else {
        hipLaunchKernelGGL(eval_program_kernel, dim3(cdiv(T, 256)), dim3(256), 0, st, a);
    }
    EZ_HIP(hipGetLastError());
    EZ_HIP(hipEventRecord(e1, st));
    if ((rc = arena_done(scratch_arena(c, st), st))) return rc;
    if (!ordered) EZ_HIP(hipStreamSynchronize(st));
    return EZKL_OK;
}

}  // namespace ezkl
