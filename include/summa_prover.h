/* summa_prover.h -- C ABI of the compiled-host `create_proof` (include/summa_prover.hpp) for the constraint system of
 * the reference's MstInclusionCircuit, exported by the same shared library as include/summa_gpu.h.
 *
 * What it replaces on the reference side [REF zk_prover/src/circuits/utils.rs]:
 *   sp_key_create      keygen_pk's prover-side forms (:76): coefficient and extended-coset forms of the fixed and
 *                      permutation columns, from their Lagrange columns (the caller has committed to them)
 *   sp_create_proof    halo2_proofs::plonk::create_proof::<KZG, ProverSHPLONK, ..> as `full_prover` (:94-101, Blake2b /
 *                      Challenge255 transcript) and `create_proof_checked` (:171-178, Keccak256Transcript) call it
 * The host part (transcripts, Fiat-Shamir scalars, the multi-open's interpolation) is C++; every data-parallel step is
 * a kernel behind include/summa_gpu.h.  Thread-safe: proofs from different host threads run side by side on the device
 * (each thread keeps its own streams and buffer pool); give every concurrent call its own `stream`.
 *
 * Conventions as in summa_gpu.h: 0 or a negative sg_status; sp_last_error() (thread-local) explains a failure. */
#ifndef SUMMA_PROVER_H
#define SUMMA_PROVER_H

#include <stddef.h>
#include <stdint.h>

#include "summa_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The constraint-system shape this prover is built for -- that of `MstInclusionConfig::configure`
 * [REF zk_prover/src/circuits/merkle_sum_tree.rs:141-207] after halo2's selector compression:
 *   3 advice columns, 1 instance column, 11 fixed columns (5 of the circuit's + 6 from its 9 selectors), one permutation
 *   argument over 6 columns in chunks of 4 (two grand products), ONE lookup with one input and one table expression
 *   (the 8-bit range check: theta-compression is the identity), constraint degree 6 (five quotient pieces on five cosets of
 *   the extended domain), 5 blinding rows + 1, rotations -1 / 0 / +1 only, instance queried at rotation 0.
 * What may vary without touching the library, because it arrives as DATA: k; the contents of every fixed / permutation column
 * (LEVELS, N_BYTES, the floor plan); the gate program and the lookup's input expression (N_CURRENCIES changes the number of sum
 * gates: any program over these columns with rotations in {-1, 0, 1} is accepted); the verifying key's digest.
 * Said once more, parameter by parameter of `MstInclusionCircuit<LEVELS, N_CURRENCIES, N_BYTES>`:
 *   LEVELS        data.  It only moves rows of the floor plan (fixed / permutation column contents) and the minimal k.
 *   N_BYTES       data.  The range check decomposes into N_BYTES lookups of ONE 8-bit table: more used rows, the same 256-entry
 *                 table column (a range table: the device-side lookup permutation applies; a table with values >= 2^16 would take
 *                 the host sort, as upstream does).
 *   N_CURRENCIES  data, with a fast path: it changes the gate program (one sum gate and one Poseidon input per currency) and the
 *                 number of instances.  Any count proves through the gate interpreter; for 1 .. 4 the library holds the lowered
 *                 programs as straight-line kernels (csrc/gates_mst_programs.inc) and the quotient numerator runs as ONE pass
 *                 (sg_quotient_numerator_cosets_dev) -- same proof bytes, fewer launches.
 * What would break it (a change of `ConstraintSystem` that needs the C++ driver in include/summa_prover.hpp rebuilt, not data):
 * another number of advice / fixed / permutation columns (SP_NUM_* below are compile-time), a second lookup or a lookup with
 * several input expressions (theta would have to be squeezed BEFORE the permuted columns are committed: the driver commits them
 * with the advice columns, summa_prover.hpp phase 2), a second instance column or an instance rotation, a rotation beyond +-1
 * (the multi-open's rotation sets are {0}, {0, 1}, {-1, 0, 1, last}: fixed), a constraint degree above 6 (more quotient pieces), a
 * permutation chunk length other than degree - 2 = 4.  sp_key_create cannot see most of these from its arguments: it checks the
 * programs' column indices and rotations and refuses what it can (SG_ERR_INVALID); the rest is the caller's contract. */
#define SP_NUM_FIXED 11
#define SP_NUM_SIGMA 6
#define SP_NUM_ADVICE 3

typedef enum sp_transcript {
  SP_TRANSCRIPT_EVM = 0,     /* Keccak256Transcript: 64-byte points, big-endian scalars (2144-byte proofs) */
  SP_TRANSCRIPT_BLAKE2B = 1  /* Blake2bWrite + Challenge255: compressed points, little-endian scalars (1632 bytes) */
} sp_transcript;

/* Proving key for 2^k rows over the SRS `srs_handle` (sg_srs_upload): d_fixed_lagrange[11] / d_sigma_lagrange[6] are
 * device columns of 2^k Montgomery Fr (copied); gates / lookup_input are the GraphEvaluator programs of the constraint
 * system (copied); vk_digest_be = the scalar `vk.hash_into` feeds the transcript, 32 bytes big-endian.
 * The gate program may read powers of the challenge y as SG_VS_CHALLENGE sources (a program that is free to fold its gate
 * polynomials in any order needs them): challenge i = sum of y^e over the i-th group of gate_challenge_exponents, the groups
 * having gate_challenge_counts[i] entries each (n_gate_challenges groups).  Work is enqueued on `stream` and complete on return. */
int sp_key_create(uint32_t k, uint64_t srs_handle, const void* const* d_fixed_lagrange, const void* const* d_sigma_lagrange,
                  const uint8_t vk_digest_be[32], const sg_graph* gates, const sg_graph* lookup_input,
                  const uint32_t* gate_challenge_exponents, const uint32_t* gate_challenge_counts, uint32_t n_gate_challenges,
                  void* stream, uint64_t* key_out);
int sp_key_destroy(uint64_t key);

/* One proof.  d_advice[3]: device columns of 2^k rows, Lagrange form; their last 6 rows are OVERWRITTEN with blinding
 * values (pass copies to keep the originals).  instances: n_instances x 32 B Montgomery Fr, the instance column's
 * values.  sanity_checks != 0: fail with SG_ERR_WITNESS if the permutation or lookup grand product does not close
 * (upstream's cargo feature of that name; without it such a witness yields a proof the verifier rejects).  A lookup
 * input outside its table is always SG_ERR_WITNESS.  proof_out: at least 2144 bytes; *proof_len = bytes written. */
int sp_create_proof(uint64_t key, void* const* d_advice, const uint8_t* instances, uint32_t n_instances, int transcript,
                    int sanity_checks, void* stream, uint8_t* proof_out, size_t proof_cap, size_t* proof_len);

const char* sp_last_error(void);

/* halo2's verify_proof::<KZG, VerifierSHPLONK, _, _, SingleStrategy> for this constraint system -- what `full_verifier`
 * (utils.rs:110-131, Blake2b flavour) and `create_proof_checked` (utils.rs:181-193, Keccak flavour, run on EVERY proof the
 * backend serves) do -- as compiled host code: transcript replay, Lagrange / instance evaluations, the constraint polynomials
 * at x, SHPLONK's scalars; the group side is one 37-point sg_msm_g1 on the device and one sg_pairing_check.
 * The verifying key is passed as data: k, N_CURRENCIES, vk.transcript_repr (32 B big-endian), the 11 fixed and 6 permutation
 * commitments (64 B each, the ABI's point format), the parameters' g2 and s_g2 (128 B each, the SRS container's format).
 * instances: n_instances x 32 B Montgomery Fr.  Returns SG_OK whenever the question could be answered: *accepted = 1 for a
 * valid proof, 0 for anything else -- wrong length, a point off the curve, an unreduced scalar, a failed pairing check;
 * other codes mean the machinery failed (no device, bad arguments), with the reason in sp_verify_last_error(). */
int sp_verify_proof(uint32_t k, uint32_t n_currencies, const uint8_t vk_digest_be[32], const uint8_t* fixed_comms,
                    const uint8_t* permutation_comms, const uint8_t g2[128], const uint8_t s_g2[128], const uint8_t* proof,
                    size_t proof_len, const uint8_t* instances, uint32_t n_instances, int transcript, int* accepted);
const char* sp_verify_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
