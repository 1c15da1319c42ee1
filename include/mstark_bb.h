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
 * every range could be); they must stay valid and unchanged until msbb_witness_destroy. Every msbb_prove on such a witness
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

#ifdef __cplusplus
}
#endif
#endif /* MSTARK_BB_H */
