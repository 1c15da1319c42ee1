// Host side of the prover: Fiat-Shamir challenger, System / ProverKey / SystemWitness mirrors and prove().
// Mirrors the reference's host orchestration (src/prover.rs:290-603, src/system.rs, src/types.rs) in C++ because
// no Rust toolchain exists in this environment; all heavy steps are launches into the kernels of msamd.h.
#pragma once
#include <memory>
#include <vector>

#include "../../include/mstark.h"
#include "b3_dev.h"
#include "msamd.h"
#include "program.h"

namespace msamd {

// ---- BLAKE3 on the host (challenger, grinding)
void blake3_host(const uint8_t* in, size_t len, uint8_t out[32]);
std::string last_error_text();  // the calling thread's ms_last_error() text (capi.hip)

// DeterministicPow<SerializingChallenger64<Goldilocks, HashChallenger<u8, Blake3, 32>>>, src/types.rs:28-81
struct Challenger {
  std::vector<uint8_t> input, output;
  explicit Challenger(const std::vector<uint8_t>& seed) : input(seed) {}
  void observe_bytes(const uint8_t* p, size_t n) {
    output.clear();
    input.insert(input.end(), p, p + n);
  }
  void observe(u64 v) {
    uint8_t b[8];
    for (int k = 0; k < 8; k++) b[k] = (uint8_t)(v >> (8 * k));
    observe_bytes(b, 8);
  }
  void observe_ext(E2 e) {
    observe(e.c0);
    observe(e.c1);
  }
  void observe_cap(const std::vector<Digest>& cap) {
    for (auto& d : cap) observe_bytes(d.b, 32);
  }
  // replace "hash(input_buffer)" by a digest computed elsewhere (device) for the pending flush
  void flush_with(const Digest& d) {
    input.assign(d.b, d.b + 32);
    output.assign(d.b, d.b + 32);
  }
  uint8_t sample_byte();
  u64 sample_u64();
  u64 sample_base();
  E2 sample_ext();
  size_t sample_bits(unsigned bits);
  u64 grind(unsigned bits);
};

struct Params {
  u64 log_blowup = 1, cap_height = 0, log_final_poly_len = 0, max_log_arity = 1, num_queries = 1, commit_pow_bits = 0,
      query_pow_bits = 0;
};

struct HCircuit {
  std::vector<PNode> nodes;
  std::vector<uint32_t> degrees, zeros;
  std::vector<std::pair<uint32_t, std::vector<uint32_t>>> lookups;
  size_t main_width = 0, pre_width = 0, pre_height = 0, num_lookups = 0, stage2_width = 0, constraint_count = 0,
         max_constraint_degree = 0, args_width = 0, lookup_prefix_len = 0;
  std::vector<u64> preprocessed;  // row-major
  DBuf<u64> d_preprocessed;       // same, on the device (witness preparation)
  DProgram prog;
  DProgram prefix_prog;           // lookup-expression prefix only (SystemWitness::from_stage_1)
  JitKernel stage2_jit;           // stage-2 terms kernel specialised to this circuit's lookups (quotient_jit.hip)
  JitKernel stage2_trace_jit;     // the same fed by the trace: lookup expressions evaluated in the kernel (host-resident witnesses)
  bool prefix_on_device = false;
  size_t quotient_degree() const {
    size_t d = (max_constraint_degree > 2 ? max_constraint_degree : 2) - 1, q = 1;
    while (q < d) q <<= 1;
    return q;
  }
};

// ProverData of one commitment: bit-reversed LDEs (column-major) + Merkle tree
struct PcsData {
  std::vector<DMat> ldes;
  DTree tree;
};

struct HSystem {
  Ctx* ctx = nullptr;
  Params params;
  std::vector<HCircuit> circuits;
  bool has_pre = false;
  std::vector<Digest> pre_commit;
  std::vector<int> pre_indices;
  PcsData pre_data;
  std::vector<uint8_t> seed;
};
std::unique_ptr<HSystem> system_from_blob(Ctx& ctx, const uint8_t* blob, size_t len);

struct HWitness {
  HSystem* sys = nullptr;
  std::vector<size_t> heights;
  std::vector<DBuf<u64>> traces;  // row-major on device, as uploaded (empty for circuits another rank computes)
  // host-resident, narrow upload by row groups (prover.hip HostUpload): the trace already transposed into bit-reversed column-major
  // storage with the first pass of the inverse transform done, group by group as the rows arrived (empty otherwise)
  std::vector<DBuf<u64>> early_evals;
  bool has_remote = false;        // some active circuit has no trace here: only prove_sharded accepts the witness
  std::vector<DLookups> lookups;
  // claims: host copy (transcript for small inputs) and device copy
  std::vector<u64> claim_offsets, claim_data;
  // A rank of a joint proof may hold only the part of the claims' data it reads (ms_witness_create_host_sliced): claim_data
  // then holds elements [claim_elem0, claim_elem0 + claim_data.size()) of claim_elems_total, claim_head the first 130 (chunk 0
  // of the claims transcript is hashed by every rank); the offsets are always complete.
  size_t claim_elem0 = 0, claim_elems_total = 0;
  std::vector<u64> claim_head;
  bool claims_partial = false;
  DBuf<u64> d_claim_offsets, d_claim_data;
  // host-resident witness (ms_witness_create_host): between proofs nothing of it lives in HBM. prove() uploads the
  // traces and the claims on the context's copy stream and runs from_stage_1 on the device, every time.
  bool host_resident = false;
  std::vector<const u64*> h_traces;                 // the caller's row-major buffers (pinned by hipHostRegister)
  std::vector<std::vector<u64>> h_mult, h_args;     // only for circuits whose lookup prefix needs the host sweep
  std::vector<void*> registered;                    // ranges this witness pinned; unpinned by the destructor
  // Narrow upload. STARK traces are mostly bytes, bits and small limbs held in 64-bit words, and the upload is the largest
  // single item of a host-resident proof (117 MB at 57 GB/s: 2.1 ms of 8.5). When every value of a trace fits 1 / 2 / 4
  // bytes (found by the validation pass at creation), prove() narrows the trace on host threads - chunk by chunk, range-
  // checked again, each chunk uploaded as soon as it is complete - and widens it on the device. A value that no longer
  // fits (the caller may rewrite its buffers between proofs) sends that proof down the plain path.
  std::vector<unsigned> pack_bytes;                 // per circuit: 0 = upload the 64-bit words as they are
  std::vector<uint8_t*> h_packed;                   // pinned staging for the narrowed rows (h * width * pack_bytes)
  // One upload of the witness: the device buffers and the events that mark their arrival. `next` is filled while a proof
  // runs when prefetching is on (ms_witness_prefetch): the following proof then finds its inputs already in HBM.
  struct Staged {
    bool valid = false, has_host_lookups = false;
    std::vector<DBuf<u64>> traces, mult, args, evals;
    std::vector<DBuf<uint8_t>> narrow;  // the narrowed rows as they arrived (kept until the upload has completed)
    DBuf<u64> claim_offsets, claim_data;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};  // traces, host lookup values, claims
    void clear() {
      valid = false;
      traces.clear();
      evals.clear();
      narrow.clear();
      mult.clear();
      args.clear();
      claim_offsets.reset();
      claim_data.reset();
    }
  };
  Staged stage[2];
  int cur = 0;             // stage[cur] feeds the running proof, stage[cur ^ 1] is the prefetched one
  bool prefetch = false;
  bool pinned = true;                               // every uploaded range is page-locked
  void pin(const void* p, size_t bytes);
  HWitness() {}
  HWitness(const HWitness&) = delete;
  HWitness& operator=(const HWitness&) = delete;
  ~HWitness();
};
// data_first / data_count: claim_data holds the elements [data_first, data_first + data_count) only (a joint proof's rank:
// claims_slice_range) and head the first min(total, 130) elements; data_count = ~0: claim_data is complete
std::unique_ptr<HWitness> witness_create_host(HSystem& sys, const u64* const* traces, const u64* heights, size_t n_claims,
                                              const u64* claim_offsets, const u64* claim_data, size_t data_first = 0,
                                              size_t data_count = ~size_t(0), const u64* head = nullptr, size_t n_head = 0);
void claims_slice_range(const HSystem& sys, const size_t* heights, size_t n_claims, const u64* offsets, size_t rank, size_t world, size_t& first,
                        size_t& count);
std::unique_ptr<HWitness> witness_create(HSystem& sys, const u64* const* traces, const u64* heights, const u64* const* mult,
                                         const u64* const* args, size_t n_claims, const u64* claim_offsets,
                                         const u64* claim_data);

// traces / claims already resident in HBM (from_stage_1 on the device); and the bench workload generated there
std::unique_ptr<HWitness> witness_from_device(HSystem& sys, std::vector<DBuf<u64>>&& traces, const std::vector<size_t>& heights,
                                              DBuf<u64>&& d_claim_offsets, DBuf<u64>&& d_claim_data, size_t n_claims, size_t claim_elems);
std::unique_ptr<HWitness> witness_u32_add_bench(HSystem& sys, size_t num_adds, u32 a0, u32 b0);

struct StageMs {
  double v[6] = {0, 0, 0, 0, 0, 0};
};
std::vector<uint8_t> prove(HSystem& sys, HWitness& w, StageMs* times);
typedef ms_comm ms_comm_t;
// the same proof computed by `comm->world` ranks (prover_sharded.inc)
std::vector<uint8_t> prove_sharded(HSystem& sys, HWitness& w, const ms_comm_t* comm, const int32_t* owners, StageMs* times);

// System::verify_multiple_claims (verifier.hip): 0 = accepted, otherwise the VerificationError code of include/mstark.h
int verify(HSystem& sys, size_t n_claims, const u64* claim_offsets, const u64* claim_data, const uint8_t* proof, size_t proof_len);

// Pcs::open / Pcs::verify on their own (prover.hip, verifier.hip)
void pcs_open_standalone(Ctx& ctx, const Params& prm, const std::vector<PcsData*>& data, const std::vector<std::vector<std::vector<E2>>>& points,
                         Challenger& ch, std::vector<E2>& opened_flat, std::vector<uint8_t>& fri_bytes);
bool pcs_verify_standalone(const Params& prm, const std::vector<std::vector<Digest>>& commits, const std::vector<std::vector<unsigned>>& log_n,
                           const std::vector<std::vector<size_t>>& widths, const std::vector<std::vector<std::vector<E2>>>& points,
                           const std::vector<E2>& opened_flat, const uint8_t* fri, size_t fri_len, Challenger& ch);

void commit_matrices(Ctx& ctx, std::vector<DMat>&& ldes, unsigned cap_height, PcsData& out);
// LookupValues::stage_2_traces of one circuit (src/lookup.rs:472-555): from the witness's lookup values, or - when the witness
// holds none for this circuit - straight from its trace with the circuit's fused kernel. Launches only; the circuit's
// total is left in *total_dev.
void stage2_circuit_async(Ctx& ctx, const HSystem& sys, const HWitness& wit, size_t ci, E2 beta, E2 gamma, u64* out_colmajor_bitrev,
                          E2* total_dev);
// the claims part of the transcript (src/prover.rs:369-373) for a device-resident witness: long lists are hashed on the device
void observe_claims(Ctx& ctx, Challenger& ch, HWitness& wit);
void field_op(Ctx& ctx, int op, const u64* a, const u64* b, size_t n, u64* out);

}  // namespace msamd
