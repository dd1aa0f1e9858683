/* ezkl_prover.h -- C ABI of the native host prover (libezkl_prover.so): halo2-shaped keygen + create_proof driving the
 * kernels of libezkl_hip.so (include/ezkl_hip.h) on resident columns.  SURVEY.md §8(f) items 3-4.
 *
 * What it replaces in the reference: halo2_proofs::plonk::{keygen_vk, keygen_pk, create_proof} as called from
 * /root/reference/src/pfsys/mod.rs:376-400 (create_keys) and :404-489 (create_proof_circuit: RNG choice :436-439,
 * instances :441-445, create_proof :456-463, transcript finalize), with the EvmTranscript of
 * /root/reference/src/execute.rs:1608-1609.  The circuit-specific part of halo2 (Circuit::configure / synthesize) stays
 * on the caller's side: the constraint system arrives as a flat description, the witness as host columns.
 *
 * The library is a pure CLIENT of include/ezkl_hip.h (it links libezkl_hip.so and calls nothing else on the GPU), so
 * it doubles as the proof that the drop-in boundary is sufficient for a complete prover.  No CPU fallback: without a
 * GPU every call that touches columns returns EZKL_ERR_NO_DEVICE.  Status codes are those of ezkl_hip.h; nothing
 * unwinds across the ABI.
 *
 * Compatibility with the zkonduit halo2 fork: the proof LAYOUT is the reference's (points 64 B BE, scalars 32 B BE) and the protocol
 * is the one the reference's own generated EVM verifier checks: tests/test_evm_verifier.py runs that bytecode
 * (/root/reference/tests/assets/wasm.code) with its verifying-key constants replaced by this library's key and it accepts the proofs
 * ezkl_prover_create_proof writes on the GPU -- same challenges, same quotient terms, same SHPLONK (halo2's query order, first-set
 * normalisation), real pairing (NOTEBOOK.md §2.1).  Not pinnable: halo2's vk digest (a hash of Debug text); the 32-byte digest here
 * binds the serialised constraint system + the commitments.  The bytes are identical to those of the Python restatement
 * ezkl_amd/plonk.py under the same randomness (tests/test_native_prover.py).
 */
#ifndef EZKL_PROVER_H
#define EZKL_PROVER_H
#include <stddef.h>
#include <stdint.h>
#include "ezkl_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- constraint system ----
 * blob (little-endian u32 words unless noted), version 2 (version 1 = the same without the fields marked [2]):
 *   magic 0x53435a45 ("EZCS"), version, k, n_advice, n_fixed, n_instance, n_challenges, advice_phase[n_advice],
 *   [2] blinding (halo2's cs.blinding_factors(); 0 = derive max(3, most queries of one advice column) + 2),
 *   [2] minimum_degree (set by halo2's chunk_lookups; 0 = none), [2] n_unblinded, unblinded advice columns[n_unblinded]
 *       (unusable rows hold Blind::default() = 1, /root/reference/src/circuit/modules/polycommit.rs:57-61),
 *   [2] n_selectors (halo2 selectors behind the fixed columns: the size of the selector section of vk.key / pk.key, which the
 *       files do not record -- halo2 re-runs configure, src/pfsys/mod.rs:627),
 *   n_nodes, nodes[n_nodes] of 48 B: {u32 op, u32 a, u32 b, u32 0, u8 constant[32] (Montgomery Fr)}
 *       op 0 CONST | 1 ADVICE(col a, rotation (i32) b) | 2 FIXED | 3 INSTANCE | 4 CHALLENGE(index a)
 *          | 5 NEG(node a) | 6 ADD(node a, node b) | 7 SUB | 8 MUL        (children precede parents)
 *   n_gates, gate_node[n_gates]                              -- the polynomials of ConstraintSystem::gates
 *   n_perm, {u32 kind (1 advice, 2 fixed, 3 instance), u32 col}[n_perm]   -- permutation::Argument columns, in order
 *   n_lookups, per lookup: n_inputs, per input {arity, node[arity]}, table {arity, node[arity]}   -- mv-lookup arguments
 *   [2] has_queries; if set: advice, fixed, instance query lists {count, {col, rotation (i32)}[count]} in halo2's order of first
 *       query (cs.advice_queries ...): the order of the evaluations in the proof (pinned on the reference's proof.json,
 *       tests/test_ezkl_circuit.py).  Without it the queries are collected from the expressions and sorted.
 * The keccak256 of the blob is bound into the vk digest (the role of halo2's vk.transcript_repr).
 */
typedef struct ezkl_prover_cs* ezkl_cs_t;
typedef struct ezkl_prover_pk* ezkl_pk_t;
int ezkl_prover_cs_parse(const void* blob, size_t len, ezkl_cs_t* out);
int ezkl_prover_cs_free(ezkl_cs_t cs);
/* out[0..8) = degree, extended_k, permutation chunk length, #z polynomials, usable rows, #advice queries,
 * #fixed queries, #instance queries */
int ezkl_prover_cs_info(ezkl_cs_t cs, uint32_t out[8]);

/* ---- multi-GPU: the MSMs of keygen / create_proof sharded by points (BASELINE configs[3]) ----
 * One process per GPU, every rank running the same deterministic prover on replicated columns.  After this call the `g` /
 * `g_lagrange` handles given to keygen / create_proof hold only points [lo, hi) of the SRS (base-set memory and MSM work divide
 * by the number of ranks); every commit batch is computed as partial sums over rows [lo, hi) and handed to `fold`, which must
 * replace the `count` 64-byte affine Montgomery points IN PLACE by their sums over all ranks (one all_gather of the partials
 * over RCCL + ezkl_hip_g1_add_affine: ezkl_amd/native.py) and return 0.  All ranks then derive the same transcript and emit the
 * same proof bytes as the unsharded prover.  An empty slice is not allowed (lo < hi <= 2^k); lo = hi = 0 switches sharding off.
 * The randomness must be the same on every rank (a shared `seed`, or an rng callback that is). */
typedef int (*ezkl_fold_fn)(void* user, void* points, uint32_t count);
int ezkl_prover_cs_set_shard(ezkl_cs_t cs, uint32_t lo, uint32_t hi, ezkl_fold_fn fold, void* user);
/* The same, over the library's own RCCL communicator (ezkl_hip_comm_init, include/ezkl_hip.h): the slice is this rank's share of the
 * 2^k points, commit batches are folded by ezkl_hip_comm_fold_points (all_gather of the 64-byte partials + group law), and with a
 * power-of-two world the quotient sweep is sharded by rows with h all_gathered in place on the device (ezkl_hip_comm_allgather_dev).
 * No callbacks, no torch: what a fork's worker process (one per GPU) calls after exchanging the unique id. */
int ezkl_prover_cs_set_shard_comm(ezkl_cs_t cs);
/* Sharding with COMPLETE base sets: tell the prover that the SRS handles passed to keygen / create_proof on this rank hold all 2^k
 * points (not the slice [lo, hi) of ezkl_prover_cs_set_shard).  A commit batch of m columns on `world` ranks is then divided by
 * columns (column i on rank i mod world: whole MSMs, no per-slice tails) and, when m < world, by 2^t point ranges inside each
 * column; a rank contributes the identity for work it does not do and the same fold callback / communicator sums the partials.
 * The advice phase then commits after its copies have landed (no PCIe / MSM overlap on that path).  Call after set_shard*. */
int ezkl_prover_cs_set_shard_full_bases(ezkl_cs_t cs, int on);
/* Optional, on top of set_shard with equal power-of-two slices: the quotient sweep sharded by ROWS.  Each rank evaluates
 * 2^ext_k / world rows of the quotient numerator (the gate program rewritten per shard: every (column, rotation) it reads is a
 * window of the resident coset column) and calls `gather`, which must make the whole device buffer `buf` (total_bytes) identical
 * on all ranks given that this rank wrote [offset, offset + bytes) -- an in-place all_gather (RCCL on device pointers;
 * ezkl_amd/native.py) -- and return 0.  gather = NULL switches it off.  sharded_sweeps counts the sweeps done this way. */
typedef int (*ezkl_gather_fn)(void* user, void* buf_dev, size_t total_bytes, size_t offset, size_t bytes);
int ezkl_prover_cs_set_sweep_gather(ezkl_cs_t cs, ezkl_gather_fn gather, void* user);
int ezkl_prover_cs_sharded_sweeps(ezkl_cs_t cs, uint64_t* out);

/* ---- multi-GPU, the full form: columns and arguments have OWNERS ("NTT by columns", the sweep by rows) ----
 * On top of set_shard* with complete base sets (set_shard_full_bases) and a power-of-two world: every witness-dependent column has
 * one owner rank -- advice column j of a phase: j mod world; lookup argument i (its compressed inputs, m_i, phi_i): i mod world;
 * permutation chunk j (z_j): j mod world.  Only the owner computes the column, its coefficient form and its extended cosets, and
 * commits it (whole MSMs; the others contribute the identity to the fold).  The extended domain is stored coset-major
 * (ezkl_hip_coeff_to_cosets_dev) and the quotient sweep is divided into max(E, world) units of rows -- whole cosets, or row ranges of
 * a coset when there are more ranks than cosets; ONE all-to-all (`exchange`: ezkl_hip_comm_alltoallv_dev's contract) moves every
 * owned column's rows of a unit (plus the few halo rows its rotations reach) to the rank that sweeps the unit; h is all_gathered
 * (set_sweep_gather); evaluations are computed by owners and all_gathered as scalars (`allgather_host`:
 * ezkl_hip_comm_allgather_host's contract); SHPLONK is linear in the polynomials, so every rank carries the partial sums over the
 * polynomials it owns through both quotients and commits its partial -- two more 64-byte folds, no polynomial ever moves.
 * The proof bytes are those of the one-GPU prover.  set_shard_comm installs the library communicator's versions by itself. */
typedef int (*ezkl_allgather_host_fn)(void* user, void* buf_host, size_t bytes_per_rank);
typedef int (*ezkl_exchange_fn)(void* user, const ezkl_comm_seg_t* sends, size_t n_sends, const ezkl_comm_seg_t* recvs, size_t n_recvs);
int ezkl_prover_cs_set_shard_exchange(ezkl_cs_t cs, ezkl_allgather_host_fn allgather_host, ezkl_exchange_fn exchange, void* user);
/* counters of the last create_proof on this rank: out[0] = witness columns this rank transformed (iNTT + cosets), out[1] = witness
 * columns in the proof, out[2] = bytes this rank received in the sweep exchange and in the two reduce-scatters of SHPLONK's partial
 * polynomials, out[3] = lookup / permutation arguments it computed */
int ezkl_prover_cs_shard_stats(ezkl_cs_t cs, uint64_t out[4]);

/* ---- one process, several GPUs: the prover group ----
 * `ezkl prove` is one process (/root/reference/src/execute.rs:1575-1627; set_device() :84-97 runs once).  A group runs the owner-mode
 * prover above on N contexts of libezkl_hip.so (ezkl_hip_init(-1): one per visible device) with one host thread per context INSIDE the
 * caller's process: commitments are folded and scalars gathered through host memory, h and the sweep's row slabs move by peer copies
 * (ezkl_hip_memcpy_peer).  No launcher, no RCCL, no worker processes; every thread reads the caller's advice columns in place.
 * group_create uses the first 2^t <= n_contexts contexts (n_contexts <= ezkl_hip_context_count()); load_srs uploads the two base sets to
 * every context and starts their window tables; keygen makes one resident key per context; create_proof returns the proof every
 * context produced (they are compared) -- the bytes of the one-GPU prover.  timings: the maximum over the contexts per stage;
 * stats (may be NULL): 4 counters per context as in ezkl_prover_cs_shard_stats.  seed = 0: a 256-bit OS-entropy key shared by all. */
typedef struct ezkl_prover_group* ezkl_group_t;
int ezkl_prover_group_create(const void* cs_blob, size_t len, int n_contexts, ezkl_group_t* out);
int ezkl_prover_group_size(ezkl_group_t group);
int ezkl_prover_group_free(ezkl_group_t group);
int ezkl_prover_group_load_srs(ezkl_group_t group, const void* g_points, const void* g_lagrange_points, size_t n);
int ezkl_prover_group_keygen(ezkl_group_t group, const void* const* fixed_values, const uint32_t* copies, size_t n_copies);
/* load_pk for the group (/root/reference/src/pfsys/mod.rs:615-636: `ezkl prove` reads pk.key): every context loads the file through
 * ezkl_prover_pk_read_file on its own thread and keeps, of the extended columns, only the cosets it sweeps (key bytes in HBM / contexts);
 * recommit != 0 recomputes the commitments under the group's SRS (ezkl_prover_pk_recommit) */
int ezkl_prover_group_pk_read_file(ezkl_group_t group, const char* path, int recommit);
int ezkl_prover_group_pk(ezkl_group_t group, int context, ezkl_pk_t* out);          /* borrowed: the key of one context (vk, verify_proof) */
int ezkl_prover_group_create_proof(ezkl_group_t group, const void* const* advice, const void* const* instances, const uint32_t* instance_lens,
                                   uint64_t seed, void* proof_out, size_t cap, size_t* proof_len, double* timings, uint64_t* stats);

/* How advice_fn hands over its columns.  Off (default): `columns[c]` points at a zeroed host buffer of 2^k x 32 B the callback fills.
 * On: `columns` arrives as an array of NULL pointers and the callback STORES, for every column c of the phase, a pointer to its own
 * host column (it may be page-locked, ezkl_hip_host_malloc) that stays valid until create_proof returns: no allocation, no copy --
 * what a fork does with the advice vectors synthesize() has just filled. */
int ezkl_prover_cs_set_advice_by_pointer(ezkl_cs_t cs, int on);

/* ---- keygen (keygen_vk + keygen_pk): fixed columns and copy constraints -> resident proving key ----
 * fixed_values: n_fixed host pointers, 2^k x 32 B Montgomery Fr each.  copies: n_copies x {colpos_a, row_a, colpos_b, row_b}
 * with colpos indexing the permutation column list.  g = the SRS in coefficient basis (ParamsKZG::g).  The cs handle is
 * borrowed and must outlive the pk. */
int ezkl_prover_keygen(ezkl_cs_t cs, ezkl_bases_t g, const void* const* fixed_values, const uint32_t* copies, size_t n_copies, ezkl_pk_t* out);
int ezkl_prover_pk_free(ezkl_pk_t pk);
/* the proving key in the raw-bytes layout of halo2's ProvingKey::{write, read} (the reference's pk.key, saved / loaded at
 * /root/reference/src/pfsys/mod.rs:615-683; layout verified on its fixture): vk prefix, l0 / l_last /
 * l_active_row, fixed values / polys / cosets, permutation values / polys / cosets.  write: EZKL_ERR_NOMEM with *len set if
 * cap is too small (a k = 20 key is GiBs).  read: the cs supplies what the file does not hold (column counts, extended_k);
 * every element is checked to be a canonical residue; columns go straight to HBM. */
/* the quotient sweep of this key, per extended row: out = [instructions, Montgomery products, column slots read, kernels].  For rooflines
 * (bench.py): one launch of the sweep runs 2^k rows of it. */
int ezkl_prover_pk_sweep_stats(ezkl_pk_t pk, uint64_t out[4]);
/* which cosets of the extended domain the key's extended columns hold and what the key occupies in HBM: out = [first coset, count, E,
 * resident key bytes].  One prover per GPU holds all E cosets; in owner mode (ezkl_prover_cs_set_shard_exchange / _comm before keygen or
 * pk_read_file) a rank computes and keeps only the cosets it sweeps. */
int ezkl_prover_pk_residency(ezkl_pk_t pk, uint64_t out[4]);
int ezkl_prover_pk_write(ezkl_pk_t pk, void* out, size_t cap, size_t* len);
int ezkl_prover_pk_read(ezkl_cs_t cs, const void* buf, size_t len, ezkl_pk_t* out);
/* load_pk for a one-shot `prove`: maps the key file and uploads only its n-row sections (fixed values, permutations); coefficient forms,
 * extended cosets and l0 / l_last / l_active are recomputed on the device instead of being read (the cosets are most of the file:
 * 4 GB of a 4.03 GB key at k = 20, ext_k = 21).  Same checks of the section headers as ezkl_prover_pk_read. */
int ezkl_prover_pk_read_file(ezkl_cs_t cs, const char* path, ezkl_pk_t* out);
/* The selector activations of the circuit (n_selectors rows of 2^k bits, bit-packed little-endian as in halo2's vk files): what
 * keygen's caller learned from synthesis; only carried into ezkl_prover_pk_write so that the key file is complete. */
int ezkl_prover_pk_set_selectors(ezkl_pk_t pk, const void* bits, size_t len);
/* Commit the key's resident fixed / permutation polynomials again under the SRS `g` (coefficient basis) and recompute the digest:
 * a key file written under one SRS (the reference's tests/assets/pk.key: the public powers of tau) proved under another. */
int ezkl_prover_pk_recommit(ezkl_pk_t pk, ezkl_bases_t g);
/* verifying key: n_fixed + n_perm affine commitments (64 B Montgomery each) and the 32-byte transcript digest
 * (Montgomery Fr).  Any output pointer may be NULL. */
int ezkl_prover_vk(ezkl_pk_t pk, void* fixed_commitments, void* permutation_commitments, void* digest);
/* Replace the digest that heads every transcript of this key by the caller's 32-byte little-endian CANONICAL scalar: a halo2 fork
 * passes vk.transcript_repr (halo2_proofs VerifyingKey::hash_into -- Blake2b of the Debug text of its pinned constraint system, which
 * only the fork can compute) so that its own verifier reads the same transcript; everything after that scalar is already the
 * reference's protocol (tests/test_evm_verifier.py).  Lost on pk_recommit / re-read (they recompute the default digest). */
int ezkl_prover_pk_set_transcript_repr(ezkl_pk_t pk, const void* repr);

/* ezkl_prover_keygen also has the sweep kernel of the circuit compiled (ezkl_hip_eval_h_prepare) and stored in the on-disk code-object cache:
 * `setup` pays hiprtc, the first `prove` -- usually another process -- loads the code object. */

/* ---- create_proof ----
 * advice: n_advice host pointers (2^k x 32 B Montgomery; rows >= usable are overwritten with blinding randomness on the
 *   device copy).  For circuits with second-phase advice pass advice_fn instead (advice may then be NULL): it is called
 *   once per phase with the challenges squeezed so far and fills columns[c] (host, 2^k x 32 B) for every column c of
 *   that phase; a non-zero return aborts with EZKL_ERR_INVALID.
 * instances: n_instance host pointers with instance_lens[i] Montgomery Fr each (hashed, not committed).
 * rng: fills n_elems x 32 B with uniform residues < r; NULL = the library's generator: ChaCha20 (the sampler of
 *   ezkl_hip_chacha20_fr_dev, uniform on [0, r)) keyed from `seed` (the reference's det-prove feature,
 *   pfsys/mod.rs:436-439) or, if seed == 0, from OS entropy; whole columns (the random polynomial) are expanded on the device.
 * proof_out / cap / proof_len: EvmTranscript bytes; EZKL_ERR_NOMEM with *proof_len set if cap is too small.
 * timings (may be NULL): 12 doubles, seconds per prover stage in the order advice_commit, lookup_m, permutation_z,
 *   lookup_phi, random_poly, intt_and_coset_ntt, quotient_sweep, h_split_commit, evaluations, shplonk, total, reserved. */
typedef int (*ezkl_advice_fn)(void* user, uint32_t phase, const void* challenges, uint32_t n_challenges, void* const* columns);
typedef void (*ezkl_rng_fn)(void* user, void* out, size_t n_elems);
int ezkl_prover_create_proof(ezkl_pk_t pk, ezkl_bases_t g, ezkl_bases_t g_lagrange, const void* const* advice, ezkl_advice_fn advice_fn,
                             void* advice_user, const void* const* instances, const uint32_t* instance_lens, ezkl_rng_fn rng, void* rng_user,
                             uint64_t seed, void* proof_out, size_t cap, size_t* proof_len, double* timings);
/* The same with a FORMAT per advice column (EZKL_COLUMN_FP / _INT64 / _INT128 of ezkl_hip.h; advice_formats[c] for column c, NULL = all
 * 32-byte Fp): every cell of an ezkl advice column is integer_rep_to_felt of an IntegerRep (/root/reference/src/fieldutils.rs:6-17), and a
 * caller that hands the integers over moves 8 or 16 bytes per cell across PCIe instead of 32 -- the columns are expanded on the device as
 * their copies land.  The proof is byte for byte the one made from the 32-byte columns.  Integer columns need caller-owned buffers: direct
 * `advice` pointers, or a callback of a constraint system set to by-pointer advice. */
int ezkl_prover_create_proof_fmt(ezkl_pk_t pk, ezkl_bases_t g, ezkl_bases_t g_lagrange, const void* const* advice, const uint8_t* advice_formats,
                                 ezkl_advice_fn advice_fn, void* advice_user, const void* const* instances, const uint32_t* instance_lens, ezkl_rng_fn rng,
                                 void* rng_user, uint64_t seed, void* proof_out, size_t cap, size_t* proof_len, double* timings);

/* ---- verify_proof: the verifier of these proofs, on the host (pairing in csrc/prover/pairing.hpp) ----
 * /root/reference/src/pfsys/mod.rs:557-590 verify_proof_circuit, and the CheckMode::SAFE self-check of create_proof_circuit (:470-480:
 * every proof is verified before it is returned).  pk supplies the constraint system and the verifying key (commitments + digest);
 * g2 / s_g2 are the 128-byte G2 elements that end the SRS file (halo2curves raw bytes: x.c0 | x.c1 | y.c0 | y.c1, Montgomery LE;
 * /root/reference/src/pfsys/srs.rs:14-16).  *accepted = 1 iff the transcript replays, the quotient identity holds at x and the SHPLONK
 * pairing check passes; a malformed proof is a rejection (0), not an error. */
int ezkl_prover_verify_proof(ezkl_pk_t pk, const void* g2, const void* s_g2, const void* proof, size_t proof_len, const void* const* instances,
                             const uint32_t* instance_lens, int* accepted);
/* The same from the verifying key ALONE, as the reference's `verify` does (settings + vk.key, /root/reference/src/execute.rs:1651): vk_buf
 * is halo2's raw-bytes vk.key ([3, k, compress] | u32 LE #fixed | commitments | selector bits; the prefix of pk.key).  Host only: no
 * device, no proving key, none of the prover's private data. */
int ezkl_prover_verify_proof_vk(ezkl_cs_t cs, const void* vk_buf, size_t vk_len, const void* g2, const void* s_g2, const void* proof, size_t proof_len,
                                const void* const* instances, const uint32_t* instance_lens, int* accepted);
/* [s] G2 for a Montgomery Fr scalar s: the `s_g2` of a test SRS (gen_srs, src/pfsys/srs.rs:13-16); s = 1 gives the generator g2. */
int ezkl_prover_g2_mul_generator(const void* scalar, void* out128);
/* keccak256 of a byte string (exposed so the transcript can be tested against the Python restatement without a GPU) */
int ezkl_prover_keccak256(const void* data, size_t len, void* out32);
/* last error text of the calling thread ("" if none) */
const char* ezkl_prover_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
