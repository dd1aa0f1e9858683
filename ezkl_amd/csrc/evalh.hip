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
#include <string.h>

namespace ezkl {

struct EvalArgs {
    const uint32_t* code;
    uint32_t n_instr;
    const fe_t* constants;
    const uint32_t* rot_off;        // rotation already scaled and reduced mod 2^ext_k
    const fe_t* const* columns;
    const fe_t* challenges;
    fe_t* interm;                   // [n_intermediates][T]
    fe_t* out;
    uint32_t ne_mask;
    uint32_t T;
};

__global__ __launch_bounds__(256) void eval_program_kernel(EvalArgs a) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t ne = a.ne_mask + 1;
    for (uint32_t r = tid; r < ne; r += a.T) {
        const fe_t prev = ld_fe(a.out + r);
        uint32_t last = 0;
        for (uint32_t ii = 0; ii < a.n_instr; ii++) {
            const uint32_t* I = a.code + 8 * (size_t)ii;
            const uint32_t op = I[0], target = I[1];
            fe_t s[2];
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const uint32_t kind = I[2 + 3 * q], idx = I[3 + 3 * q], rot = I[4 + 3 * q];
                if (q == 1 && (op == EZKL_OP_SQUARE || op == EZKL_OP_DOUBLE || op == EZKL_OP_NEGATE || op == EZKL_OP_STORE)) break;
                switch (kind) {
                case EZKL_SRC_CONST: s[q] = ld_fe(a.constants + idx); break;
                case EZKL_SRC_INTERMEDIATE: s[q] = ld_fe(a.interm + (size_t)idx * a.T + tid); break;
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
            default: t = Fr::add(Fr::mul(ld_fe(a.interm + (size_t)target * a.T + tid), s[1]), s[0]); break;
            }
            st_fe(a.interm + (size_t)target * a.T + tid, t);
            last = target;
        }
        if (a.n_instr) st_fe(a.out + r, ld_fe(a.interm + (size_t)last * a.T + tid));
    }
}

int eval_program(Ctx* c, hipStream_t st, const ezkl_program_t* p, fe_t* out) {
    if (p->ext_k > 28 || p->k > p->ext_k) return EZKL_ERR_INVALID;
    if (p->n_instr == 0) return EZKL_OK;
    const size_t ne = (size_t)1 << p->ext_k;
    // validate the program before it touches the device
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
    size_t T = (size_t)c->num_cus * 256 * 4;
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
    size_t total = o_int + al((size_t)p->n_intermediates * T * 32);
    uint8_t* S = nullptr;
    int rc = scratch_reserve(c, total, (void**)&S);
    if (rc) return rc;
    EZ_HIP(hipMemcpyAsync(S + o_code, p->code, (size_t)p->n_instr * 32, hipMemcpyHostToDevice, st));
    if (p->n_constants) EZ_HIP(hipMemcpyAsync(S + o_const, p->constants, (size_t)p->n_constants * 32, hipMemcpyHostToDevice, st));
    EZ_HIP(hipMemcpyAsync(S + o_rot, rot.data(), rot.size() * 4, hipMemcpyHostToDevice, st));
    if (p->n_columns) EZ_HIP(hipMemcpyAsync(S + o_cols, p->columns, (size_t)p->n_columns * 8, hipMemcpyHostToDevice, st));
    if (p->n_challenges) EZ_HIP(hipMemcpyAsync(S + o_chal, p->challenges, (size_t)p->n_challenges * 32, hipMemcpyHostToDevice, st));
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
    hipEvent_t e0, e1;
    if ((rc = ev_pair(c, "eval_h", &e0, &e1))) return rc;
    EZ_HIP(hipEventRecord(e0, st));
    hipLaunchKernelGGL(eval_program_kernel, dim3(cdiv(T, 256)), dim3(256), 0, st, a);
    EZ_HIP(hipGetLastError());
    EZ_HIP(hipEventRecord(e1, st));
    EZ_HIP(hipStreamSynchronize(st));   // the host-side program arrays are borrowed only for the call
    return EZKL_OK;
}

}  // namespace ezkl
