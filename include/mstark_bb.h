/* mstark_bb.h — C ABI of the MI355X (gfx950) prover for multi-stark's SECOND configuration: BabyBear, degree-4
 * binomial extension, Poseidon2 sponge / compression, DuplexChallenger — the StarkGenericConfig instantiated in
 * /root/reference/src/test_circuits/baby_bear_config.rs:28-127 (BASELINE config 4). Same conventions as mstark.h
 * (which this header builds on: ms_ctx, status codes, ms_last_error), with these differences:
 *   - BabyBear elements cross the ABI as canonical u32; an Ext4 value is four consecutive u32 (c0..c3); a digest is
 *     eight u32 ([BabyBear; 8]);
 *   - the proof bytes carry every field element the way p3-monty-31 serialises it (its Montgomery word, 4 bytes LE);
 *   - the system blob starts with the magic "MSYB01" and carries, after the seven parameters, the 141 round constants
 *     of Poseidon2BabyBear<16> (8 x 16 external, then 13 internal, canonical). The reference draws them from
 *     rand::SmallRng::seed_from_u64(42) (baby_bear_config.rs:54-55); that stream cannot be reproduced without the crate,
 *     so a maintainer passes `perm`'s constants in.
 *
 * What each entry point replaces in /root/reference (SC = BabyBearPoseidon2Config):
 *   msbb_system_*    System::<SC>::new + ProverKey                   src/system.rs:115-203
 *   msbb_witness_*   SystemWitness::from_stage_1                      src/system.rs:225-328
 *   msbb_prove       System::<SC>::prove_multiple_claims              src/prover.rs:290-603
 *   msbb_dft_batch / msbb_coset_lde_batch    Radix2DitParallel<BabyBear>   baby_bear_config.rs:38; src/prover.rs:350,419,650,716
 *   msbb_mmcs_*      MerkleTreeMmcs<.., PaddingFreeSponge<Perm,16,8,8>, TruncatedPermutation<Perm,2,8,16>, 2, 8>   baby_bear_config.rs:30-33
 *   msbb_poseidon2_permute   Poseidon2BabyBear<16>::permute           baby_bear_config.rs:29
 *   msbb_verify      System::<SC>::verify_multiple_claims             src/verifier.rs:208-532
 */
#ifndef MSTARK_BB_H
#define MSTARK_BB_H

#include "mstark.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msbb_system msbb_system;   /* System<BabyBearPoseidon2Config> + ProverKey */
typedef struct msbb_witness msbb_witness; /* SystemWitness + claims, resident in HBM */
typedef struct msbb_mmcs msbb_mmcs;       /* ProverData of one commitment */

int32_t msbb_system_create(ms_ctx* ctx, const uint8_t* blob, size_t len, msbb_system** out);
void msbb_system_destroy(msbb_system* sys);
/* preprocessed commitment: n_digests * 8 canonical words */
int32_t msbb_system_preprocessed_commit(const msbb_system* sys, uint32_t* out, size_t cap_words, size_t* n_digests);
/* the nine numbers of ms_system_circuit_info */
int32_t msbb_system_circuit_info(const msbb_system* sys, size_t circuit, uint64_t out9[9]);

/* traces[i]: heights[i] x main_width_i row-major canonical u32 (height 0 = inactive circuit). The lookup values of
 * SystemWitness::from_stage_1 (src/system.rs:244-328) are computed on the device. Claims: offsets (n_claims + 1) into
 * claim_data. */
int32_t msbb_witness_create(msbb_system* sys, const uint32_t* const* traces, const uint64_t* heights, size_t n_claims,
                            const uint64_t* claim_offsets, const uint32_t* claim_data, msbb_witness** out);
/* A SystemWitness that STAYS in host memory - what the reference's prove() is handed (src/prover.rs:290-295; the setup closure
 * of a criterion bench builds it). Values are validated and the caller's trace buffers page-locked here (*pinned = 1 when
 * every range could be; otherwise the uploads go through a page-locked bounce buffer; the locks are counted per range as for
 * ms_witness_create_host); they must stay valid and unchanged until msbb_witness_destroy. Every msbb_prove on such a witness
 * uploads traces and claims, runs from_stage_1 on the device and releases the device copies: its wall time is the reference's
 * timed region (witness in host memory at the start, proof bytes in host memory at the end). */
int32_t msbb_witness_create_host(msbb_system* sys, const uint32_t* const* traces, const uint64_t* heights, size_t n_claims,
                                 const uint64_t* claim_offsets, const uint32_t* claim_data, int32_t* pinned /* nullable */,
                                 msbb_witness** out);
void msbb_witness_destroy(msbb_witness* w);

/* Writes Proof::to_bytes (src/prover.rs:241-248). stage_ms (optional, 6 doubles) as for ms_prove. */
int32_t msbb_prove(msbb_system* sys, msbb_witness* w, uint8_t* proof_out, size_t cap, size_t* proof_len, double* stage_ms);

/* System::<SC>::verify_multiple_claims (src/verifier.rs:208-532, shape checks :536-695) on the bytes msbb_prove wrote;
 * *verdict as for ms_verify (MS_VERDICT_*). Host code. */
int32_t msbb_verify(msbb_system* sys, size_t n_claims, const uint64_t* claim_offsets, const uint32_t* claim_data,
                    const uint8_t* proof, size_t proof_len, int32_t* verdict);

/* ---- PCS-level entry points (host buffers in and out, canonical u32) */
/* the permutation used by msbb_poseidon2_permute / msbb_mmcs_commit: 141 canonical round constants */
int32_t msbb_set_poseidon2(ms_ctx* ctx, const uint32_t* constants141);
/* n states of 16 words, permuted in place */
int32_t msbb_poseidon2_permute(ms_ctx* ctx, uint32_t* states, size_t n);
int32_t msbb_dft_batch(ms_ctx* ctx, const uint32_t* in, size_t h, size_t w, int32_t inverse, uint32_t* out);
/* coset_lde_batch(evals, log_blowup, GENERATOR = 31).bit_reverse_rows() */
int32_t msbb_coset_lde_batch(ms_ctx* ctx, const uint32_t* in, size_t h, size_t w, uint32_t log_blowup, uint32_t* out);
/* cap_out: (8 << cap_height) words */
int32_t msbb_mmcs_commit(ms_ctx* ctx, size_t n, const uint32_t* const* mats, const uint64_t* heights, const uint64_t* widths,
                         uint32_t cap_height, uint32_t* cap_out, msbb_mmcs** out);
/* opened rows concatenated in matrix order; siblings bottom-up, 8 words each */
int32_t msbb_mmcs_open(msbb_mmcs* m, size_t index, uint32_t* vals_out, uint32_t* proof_out, size_t* n_siblings);
void msbb_mmcs_destroy(msbb_mmcs* m);
/* op 0 add, 1 sub, 2 mul, 3 inverse(a), 4 ext4 mul (quads), 5 ext4 inverse (quads) */
int32_t msbb_field_op(ms_ctx* ctx, int32_t op, const uint32_t* a, const uint32_t* b, size_t n, uint32_t* out);

/* ---- Level 2: the prover's steps on DEVICE HANDLES for this configuration, the counterpart of include/mstark.h's Level 2 - a
 * host that keeps the reference's own prover loop (src/prover.rs:290-603) calls the device once per step, and only
 * commitments, accumulators and challenges cross the boundary. Field elements are canonical u32, extension elements four
 * of them, digests eight (as everywhere in this header; the proof bytes hold Montgomery words, src/prover.rs:241-248).
 * tests/test_gpu_bb_level2.py drives the loop from Python and obtains the bytes msbb_prove writes.
 *   msbb_challenger_create          config.initialise_challenger()                       baby_bear_config.rs:108-114
 *   msbb_challenger_observe / _observe_digests / _sample_ext / _sample_bits              DuplexChallenger, baby_bear_config.rs:37
 *   msbb_challenger_observe_claims  the claims absorbed by the transcript                src/prover.rs:369-373
 *   msbb_witness_commit_stage1      pcs.commit(stage-1 traces)                           src/prover.rs:338-350
 *   msbb_witness_claims_accumulator the initial accumulator                              src/prover.rs:382-387
 *   msbb_stage2_build               LookupValues::stage_2_traces -> evaluation handles   src/prover.rs:400, src/lookup.rs:472-555
 *   msbb_pcs_commit_traces          pcs.commit(stage-2 traces) on those handles          src/prover.rs:414-419
 *   msbb_quotient                   quotient_values + slices + LDE -> LDE handle         src/prover.rs:459-468,483,511-517
 *   msbb_pcs_commit_ldes            pcs.commit_ldes                                      src/prover.rs:526
 *   msbb_system_preprocessed_mmcs   the ProverKey's preprocessed prover data (a view)    src/system.rs:190-195
 *   msbb_pcs_open                   pcs.open: opened values + the FriProof bytes         src/prover.rs:580
 * Matrices of a commitment are indexed by ACTIVE position. A handle consumed by a commit call keeps existing but is empty.
 * The commitment parameters (log_blowup, cap_height) and the FRI parameters are the system's. Host-resident witnesses are
 * not accepted. msbb_pcs_open: n_points has one entry per matrix (rounds flattened, at most two points per matrix), points
 * four words per point; opened values come back in round -> matrix -> point -> column order, four words each;
 * MS_ERR_BUFFER with *fri_len = the needed size when a capacity is too small (the challenger has then been advanced). */
typedef struct msbb_challenger msbb_challenger;
typedef struct msbb_trace msbb_trace;
int32_t msbb_challenger_create(msbb_system* sys, msbb_challenger** out);
void msbb_challenger_destroy(msbb_challenger* ch);
int32_t msbb_challenger_observe(msbb_challenger* ch, const uint32_t* elems, size_t n);
int32_t msbb_challenger_observe_digests(msbb_challenger* ch, const uint32_t* digests, size_t n);
int32_t msbb_challenger_sample_ext(msbb_challenger* ch, uint32_t out4[4]);
int32_t msbb_challenger_sample_bits(msbb_challenger* ch, uint32_t bits, uint64_t* out);
int32_t msbb_challenger_observe_claims(msbb_challenger* ch, msbb_witness* w);
void msbb_trace_destroy(msbb_trace* t);
int32_t msbb_trace_info(const msbb_trace* t, uint64_t out3[3]); /* height, width, kind (0 evaluations, 1 LDE) */
int32_t msbb_system_preprocessed_mmcs(msbb_system* sys, msbb_mmcs** out); /* *out = NULL when the system has no preprocessed trace */
int32_t msbb_witness_commit_stage1(msbb_witness* w, uint32_t* cap_out, msbb_mmcs** out);
int32_t msbb_witness_claims_accumulator(msbb_witness* w, const uint32_t beta[4], const uint32_t gamma[4], uint32_t acc_out[4]);
/* accs_out: 4 words per active circuit (the accumulator after each circuit); traces_out: one handle per active circuit */
int32_t msbb_stage2_build(msbb_witness* w, const uint32_t beta[4], const uint32_t gamma[4], const uint32_t acc_in[4], uint32_t* accs_out,
                          msbb_trace** traces_out);
int32_t msbb_pcs_commit_traces(msbb_system* sys, size_t n, msbb_trace* const* evals, uint32_t* cap_out, msbb_mmcs** out);
/* publics16 = [beta, gamma, acc_in, acc_out] (src/lookup.rs:78-84) */
int32_t msbb_quotient(msbb_system* sys, size_t circuit, uint32_t log_n, msbb_mmcs* s1, size_t s1_idx, msbb_mmcs* s2, size_t s2_idx,
                      const uint32_t publics16[16], const uint32_t alpha[4], msbb_trace** q_lde_out);
int32_t msbb_pcs_commit_ldes(msbb_system* sys, size_t n, msbb_trace* const* ldes, uint32_t* cap_out, msbb_mmcs** out);
int32_t msbb_pcs_open(msbb_system* sys, size_t n_rounds, msbb_mmcs* const* rounds, const uint64_t* n_points, const uint32_t* points,
                      msbb_challenger* ch, uint32_t* opened_out, size_t opened_cap_words, uint8_t* fri_out, size_t fri_cap, size_t* fri_len);

#ifdef __cplusplus
}
#endif
#endif /* MSTARK_BB_H */
