// Internal header of the BabyBear / Poseidon2 path (the reference's second StarkGenericConfig,
// /root/reference/src/test_circuits/baby_bear_config.rs): device matrices, Merkle trees and kernel launchers.
// Layout in HBM: every matrix is column-major u32 in Montgomery form (a column = one polynomial, contiguous, so the
// transforms stream whole columns and a leaf-hash thread reads row r of every column with unit stride across the wave);
// extension-field vectors (FRI layers, reduced openings, stage-2 terms) are arrays of E4 (16 bytes).
#pragma once
#include <memory>
#include <vector>

#include "bb_dev.h"
#include "msamd.h"
#include "program.h"

namespace msbb {

using msamd::Ctx;
using msamd::DBuf;

struct BMat {
  DBuf<u32> buf;
  size_t h = 0, w = 0, ld = 0;  // element (r, c) at buf.p[c * ld + r]
  u32* col(size_t c) { return buf.p + c * ld; }
  const u32* col(size_t c) const { return buf.p + c * ld; }
};
inline BMat bmat(Ctx& ctx, size_t h, size_t w) {
  BMat m;
  m.h = h, m.w = w, m.ld = h;
  m.buf = DBuf<u32>(ctx, h * w);
  return m;
}

// MerkleTreeMmcs<Packing, Packing, PaddingFreeSponge<Perm,16,8,8>, TruncatedPermutation<Perm,2,8,16>, 2, 8>
struct BTree {
  std::vector<DBuf<Digest8>> layers;  // layers[0] = leaf layer
  std::vector<size_t> sizes;
  unsigned cap_height = 0;
  size_t cap_layer() const { return layers.size() - 1 - std::min<size_t>(cap_height, layers.size() - 1); }
};
struct BPcsData {
  std::vector<BMat> ldes;  // input order
  BTree tree;
};

// ---- layout / conversion
void bb_upload_rows(Ctx& ctx, const u32* host_rowmajor_canonical, size_t h, size_t w, BMat& out);
void bb_upload_rows_async(Ctx& ctx, const u32* host_rowmajor_canonical, size_t h, size_t w, BMat& out);  // queued, no synchronisation
void bb_download_rows(Ctx& ctx, const BMat& m, bool bitrev_rows, u32* host_rowmajor_canonical);
// ---- transforms (forward DIF: natural in, bit-reversed out, in place on every column)
void bb_dif(Ctx& ctx, u32* data, size_t ld, unsigned log_n, size_t ncols);
// coset LDE: evaluations on H_n (natural) -> evaluations on GENERATOR * H_{n << lb}, stored bit-reversed
void bb_coset_lde(Ctx& ctx, const BMat& evals, unsigned log_blowup, BMat& out);
// quotient evaluations (natural order on GENERATOR * H_{nq}, D = 4 columns) -> committed LDE ((n << lb) x (4 q))
void bb_quotient_lde(Ctx& ctx, BMat& q_evals, unsigned log_n, unsigned log_q, unsigned log_blowup, BMat& out);
// ---- hashing
void bb_commit(Ctx& ctx, const Poseidon2* d_perm, std::vector<BMat>&& ldes, unsigned cap_height, BPcsData& out);
void bb_permute_batch(Ctx& ctx, const Poseidon2* d_perm, u32* d_states, size_t n);
// Merkle tree over a vector of E4 pairs (FRI layer: row i = (v[2i], v[2i+1]) flattened to 8 base columns)
struct DevChallenger;
struct FriBeta;
// with d_ch the launch that produces the root also runs the round's challenger step (observe the cap, sample beta -> *d_beta_out)
// log_arity: the round commits rows of 2^log_arity values (FriParameters::max_log_arity, src/types.rs:189-190); the same bound as
// the Goldilocks path's (msamd.h FRI_MAX_LOG_ARITY), although the sponge itself has none
static const unsigned BB_FRI_MAX_LOG_ARITY = 6;
void bb_commit_pairs(Ctx& ctx, const Poseidon2* d_perm, const E4* d_vec, size_t rows, unsigned cap_height, BTree& out,
                     DevChallenger* d_ch = nullptr, FriBeta* d_beta_out = nullptr, unsigned log_arity = 1);
// a whole commit-phase round in one launch (fold with the previous round's beta, leaf digests, tree, challenger step)
bool bb_fri_round_fusable(size_t rows, unsigned cap_height);
void bb_fri_round_fused(Ctx& ctx, const Poseidon2* d_perm, const E4* cur, size_t rows, const FriBeta* prev, const E4* roll_in, E4* out,
                        BTree& t, DevChallenger* d_ch, FriBeta* d_beta_out);

// ---- node programs (graph::Node, src/graph.rs:35-46) on the device
struct BProgram {
  DBuf<u32> kind, a, b;  // kind | source << 8 | offset << 16; constants in Montgomery form
  size_t n = 0;
};
void bb_build_program(Ctx& ctx, const std::vector<msamd::PNode>& nodes, BProgram& out);

struct BLookupsDev {
  DBuf<u32> mult, arg_off, args;  // node ids; arg_off has L + 1 entries
  size_t L = 0;
};
// LookupValues::stage_2_traces for one circuit (src/lookup.rs:472-555): the stage-2 matrix (n x 4 max(L,1)) and the
// circuit's local total (added to the running accumulator by the caller)
void bb_stage2(Ctx& ctx, const BProgram& prog, size_t prefix_len, const BLookupsDev& lk, const BMat& trace, const BMat* pre, E4 beta,
               E4 gamma, BMat& out, E4* total);
// sum over the claims of 1 / (beta + fingerprint(gamma, claim)) (src/prover.rs:382-387); data in Montgomery form
E4 bb_claims_accumulator(Ctx& ctx, const u32* d_data_monty, const u64* d_offs, size_t n, E4 beta, E4 gamma);
// proof-of-work search of the duplex challenger on the device: smallest canonical witness
u32 bb_grind(Ctx& ctx, const Poseidon2* d_perm, const u32* state16, const u32* pending, unsigned n_pending, unsigned bits);
// quotient_values (src/prover.rs:756-962) on the quotient domain; q_evals: (n q) x 4 in natural order
struct BQuotientIn {
  const BProgram* prog;
  const BLookupsDev* lk;
  const u32* d_zeros;
  size_t n_zeros, constraint_count;
  const BMat *pre, *s1, *s2;  // LDEs (bit-reversed storage); pre may be null
  unsigned log_n, log_q, log_blowup;
  E4 publics[4];  // beta, gamma, acc_initial, acc_final
  E4 alpha;
  const msamd::JitKernel* jit = nullptr;  // the circuit's compiled kernel (quotient_jit.hip), or none: the interpreter runs
};
void bb_quotient(Ctx& ctx, const BQuotientIn& in, BMat& q_evals);

// ---- opening
// 1 / (z - x_i) and x_i / (z - x_i) for the first h storage rows of the bit-reversed coset GENERATOR * H
void bb_inv_denoms(Ctx& ctx, E4 z, unsigned log_h, size_t count, E4* d_inv, E4* d_wgt);
// sum_i wgt[i] * m[i][c] over the first h rows, per column (unscaled): partial sums per row block on the device
// (bb_bary_partials(w, h) values), read back once for all matrices and points, then summed on the host
size_t bb_bary_partials(size_t w, size_t h);
void bb_bary_launch(Ctx& ctx, const BMat& m, size_t h, const E4* d_wgt, E4* d_part);
void bb_bary_finish(const E4* h_part, size_t w, size_t h, std::vector<E4>& sums);
// ro[i] += sum_p dinv_p[i] * (K_p - off_p * sum_c apow[c] m[i][c])
void bb_deep(Ctx& ctx, const BMat& m, const E4* d_apow, int npoints, const E4* const* d_inv, const E4* K, const E4* off, E4* d_ro);
void bb_fri_fold(Ctx& ctx, const E4* cur, size_t rows_out, E4 beta, const E4* roll_in, E4* out);
// the commit-phase transcript on the device (no proof of work): duplex challenger state, one step per round
struct DevChallenger {
  u32 state[16], input[8], n_in, n_out;  // the output buffer is state[..n_out], popped from the back
};
struct FriBeta {
  E4 beta, half_beta, beta2;
};
void bb_fri_challenge(Ctx& ctx, DevChallenger* d_ch, const Digest8* d_cap, size_t n_cap, const Poseidon2* d_perm, FriBeta* d_out);
void bb_fri_fold_dev(Ctx& ctx, const E4* cur, size_t rows_out, const FriBeta* d_beta, const E4* roll_in, E4* out, unsigned squarings = 0);
// gather scattered words into one buffer: out[dst + k] = src[k * stride], k < n
struct GatherSeg {
  const u32* src;
  u32 dst, n, stride;
};
void bb_gather(Ctx& ctx, const std::vector<GatherSeg>& segs, std::vector<u32>& out);
void bb_field_op(Ctx& ctx, int op, const u32* a, const u32* b, size_t n, u32* out);

}  // namespace msbb
