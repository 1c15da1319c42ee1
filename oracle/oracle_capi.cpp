// ORACLE — test infrastructure only. extern "C" surface for ctypes (tests/, smoke(), bench.py cpu_baseline).
#include <cstring>
#include <string>

#include "blake3_ref.hpp"
#include "ms_oracle.hpp"

using namespace mso;

static thread_local std::string g_err;
#define TRY try {
#define CATCH                       \
  }                                 \
  catch (const std::exception& e) { \
    g_err = e.what();               \
    return -1;                      \
  }                                 \
  catch (...) {                     \
    g_err = "unknown error";        \
    return -1;                      \
  }

static std::vector<std::vector<u64>> unpack_claims(size_t n, const u64* offsets, const u64* data) {
  std::vector<std::vector<u64>> c(n);
  for (size_t i = 0; i < n; i++) c[i].assign(data + offsets[i], data + offsets[i + 1]);
  return c;
}
static Mat mat_from(const u64* p, size_t h, size_t w) {
  Mat m(h, w);
  if (h * w) memcpy(m.v.data(), p, h * w * 8);
  return m;
}

extern "C" {

const char* mso_last_error() { return g_err.c_str(); }
#ifdef _OPENMP
}
#include <omp.h>
extern "C" {
void mso_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int mso_max_threads() { return omp_get_max_threads(); }
#else
void mso_set_threads(int) {}
int mso_max_threads() { return 1; }
#endif

// ---- field / hash primitives (KAT pinning)
u64 mso_gl_mul(u64 a, u64 b) { return f_mul(a, b); }
u64 mso_gl_add(u64 a, u64 b) { return f_add(a, b); }
u64 mso_gl_sub(u64 a, u64 b) { return f_sub(a, b); }
u64 mso_gl_inv(u64 a) { return f_inv(a); }
u64 mso_gl_two_adic_generator(unsigned bits) { return f_two_adic_generator(bits); }
// extension elements cross this surface as EXT_D consecutive u64 (2 for Goldilocks, 4 for BabyBear)
unsigned mso_ext_degree() { return EXT_D; }
u64 mso_field_order() { return F_P; }
void mso_e2_mul(const u64* a, const u64* b, u64* o) { ef_to(ef_mul(ef_from(a), ef_from(b)), o); }
void mso_e2_inv(const u64* a, u64* o) { ef_to(ef_inv(ef_from(a)), o); }
#ifdef MSO_BABYBEAR
// 8*16 external then 13 internal round constants, canonical
int mso_set_poseidon2(const u64* k141) {
  Poseidon2Constants& k = poseidon2_constants();
  for (int i = 0; i < 141; i++)
    if (k141[i] >= F_P) return -1;
  for (int r = 0; r < 8; r++)
    for (int i = 0; i < 16; i++) k.external[r][i] = k141[16 * r + i];
  for (int r = 0; r < 13; r++) k.internal[r] = k141[128 + r];
  return 0;
}
void mso_poseidon2_permute(u64* state16) { poseidon2_permute(state16); }
u64 mso_to_wire(u64 x) { return bb_to_wire(x); }
#endif
void mso_b3_g(uint32_t* v4, uint32_t mx, uint32_t my) {
  uint32_t v[16] = {0};
  v[0] = v4[0];
  v[4] = v4[1];
  v[8] = v4[2];
  v[12] = v4[3];
  b3_g(v, 0, 4, 8, 12, mx, my);
  v4[0] = v[0];
  v4[1] = v[4];
  v4[2] = v[8];
  v4[3] = v[12];
}
// the reference KAT's 32-word layout: state[0..16] = v, state[16..32] = message; output 16 words
void mso_b3_rounds_kat(const uint32_t* state_in32, uint32_t* out16) {
  uint32_t v[16];
  memcpy(v, state_in32, 64);
  b3_rounds(v, state_in32 + 16);
  for (int i = 0; i < 8; i++) {
    out16[i] = v[i] ^ v[i + 8];
    out16[i + 8] = v[i + 8] ^ state_in32[i];
  }
}
void mso_hash_bytes(const uint8_t* p, size_t n, uint8_t* out32) { blake3_hash(p, n, out32); }
void mso_hash_elems(const u64* e, size_t n, uint8_t* out32) {
  Digest d = hash_elems(e, n);
  memcpy(out32, d.b, 32);
}
void mso_compress2(const uint8_t* l, const uint8_t* r, uint8_t* out32) {
  Digest a, b;
  memcpy(a.b, l, 32);
  memcpy(b.b, r, 32);
  Digest d = compress2(a, b);
  memcpy(out32, d.b, 32);
}

// ---- DFT family (row-major in / row-major out)
int mso_dft_batch(const u64* in, size_t h, size_t w, int inverse, u64* out) {
  TRY Mat m = mat_from(in, h, w);
  Mat o = inverse ? idft_batch(m) : dft_batch(m);
  memcpy(out, o.v.data(), h * w * 8);
  return 0;
  CATCH
}
int mso_coset_lde_bitrev(const u64* in, size_t h, size_t w, unsigned log_blowup, u64 shift, u64* out) {
  TRY Mat o = coset_lde_bitrev(mat_from(in, h, w), log_blowup, shift);
  memcpy(out, o.v.data(), o.v.size() * 8);
  return 0;
  CATCH
}
int mso_shifted_quotient_slices(const u64* in, size_t h, size_t w, size_t qdeg, u64* out) {
  TRY Mat o = shifted_quotient_slices(mat_from(in, h, w), F_GENERATOR, qdeg);
  memcpy(out, o.v.data(), o.v.size() * 8);
  return 0;
  CATCH
}
int mso_lde_from_shifted_coefficients(const u64* in, size_t h, size_t w, unsigned log_blowup, u64* out) {
  TRY Mat o = lde_from_shifted_coefficients(mat_from(in, h, w), log_blowup);
  memcpy(out, o.v.data(), o.v.size() * 8);
  return 0;
  CATCH
}

// ---- Merkle MMCS
void* mso_mmcs_commit(size_t n, const u64* const* mats, const u64* heights, const u64* widths, unsigned cap_height,
                      uint8_t* cap_out) {
  try {
    std::vector<Mat> ms;
    for (size_t i = 0; i < n; i++) ms.push_back(mat_from(mats[i], heights[i], widths[i]));
    MerkleTree* t = new MerkleTree();
    mmcs_commit(std::move(ms), cap_height, *t);
    auto cap = t->cap();
    for (size_t i = 0; i < cap.size(); i++) memcpy(cap_out + 32 * i, cap[i].b, 32);
    return t;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void mso_mmcs_free(void* t) { delete (MerkleTree*)t; }
// opened values concatenated in matrix order into vals_out; siblings into proof_out; returns #siblings
int mso_mmcs_open(void* tp, size_t index, u64* vals_out, uint8_t* proof_out) {
  TRY MerkleTree* t = (MerkleTree*)tp;
  BatchOpening bo = mmcs_open_batch(*t, index);
  size_t k = 0;
  for (auto& row : bo.opened_values)
    for (u64 x : row) vals_out[k++] = x;
  for (size_t i = 0; i < bo.proof.size(); i++) memcpy(proof_out + 32 * i, bo.proof[i].b, 32);
  return (int)bo.proof.size();
  CATCH
}
int mso_mmcs_verify(const uint8_t* cap, size_t ncap, size_t n, const u64* heights, const u64* widths, size_t index,
                    const u64* vals, const uint8_t* proof, size_t nproof) {
  TRY std::vector<Digest> c(ncap);
  for (size_t i = 0; i < ncap; i++) memcpy(c[i].b, cap + 32 * i, 32);
  std::vector<Dim> dims;
  BatchOpening bo;
  size_t k = 0;
  for (size_t i = 0; i < n; i++) {
    dims.push_back(Dim{(size_t)widths[i], (size_t)heights[i]});
    bo.opened_values.emplace_back(vals + k, vals + k + widths[i]);
    k += widths[i];
  }
  bo.proof.resize(nproof);
  for (size_t i = 0; i < nproof; i++) memcpy(bo.proof[i].b, proof + 32 * i, 32);
  return mmcs_verify_batch(c, dims, index, bo) ? 1 : 0;
  CATCH
}

// ---- challenger
#ifndef MSO_BABYBEAR
void* mso_challenger_new(const uint8_t* seed, size_t n) { return new Challenger(std::vector<uint8_t>(seed, seed + n)); }
void mso_challenger_observe_bytes(void* c, const uint8_t* p, size_t n) { ((Challenger*)c)->observe_bytes(p, n); }
#else
// the seed bytes are observed one field element each (the way baby_bear_config.rs:72-75 feeds its tag)
void* mso_challenger_new(const uint8_t* seed, size_t n) { return new Challenger(std::vector<u64>(seed, seed + n)); }
void mso_challenger_observe_bytes(void* c, const uint8_t* p, size_t n) {
  for (size_t i = 0; i < n; i++) ((Challenger*)c)->observe(p[i]);
}
#endif
void mso_challenger_free(void* c) { delete (Challenger*)c; }
void mso_challenger_observe(void* c, u64 x) { ((Challenger*)c)->observe(x); }
void mso_challenger_sample_ext(void* c, u64* out2) {
  ef_to(((Challenger*)c)->sample_ext(), out2);
}
u64 mso_challenger_sample_bits(void* c, unsigned bits) { return ((Challenger*)c)->sample_bits(bits); }
u64 mso_challenger_grind(void* c, unsigned bits) { return ((Challenger*)c)->grind(bits); }

// ---- Pcs::open / Pcs::verify on their own. params7 = the seven CommitmentParameters / FriParameters words; n_points has one
// entry per matrix (rounds flattened), points holds EXT_D words each; opened values are written / read in
// round -> matrix -> point -> column order, EXT_D words each. mso_pcs_open returns the FRI proof length (-needed if the
// buffer is too small, -1 on error); mso_pcs_verify returns 1 = accepted, 0 = rejected, -1 = error.
static Params params_from(const u64* p7) {
  Params p;
  p.log_blowup = p7[0], p.cap_height = p7[1], p.log_final_poly_len = p7[2], p.max_log_arity = p7[3], p.num_queries = p7[4];
  p.commit_pow_bits = p7[5], p.query_pow_bits = p7[6];
  if (p.max_log_arity < 1 || p.max_log_arity > 16) throw std::runtime_error("max_log_arity out of range (1..16)");
  return p;
}
long mso_pcs_open(const u64* params7, size_t n_rounds, void* const* mmcs, const u64* n_points, const u64* points, void* challenger,
                  u64* opened_out, uint8_t* fri_out, size_t fri_cap) {
  try {
    Params prm = params_from(params7);
    std::vector<PcsOpenRound> rounds;
    size_t mk = 0, pk = 0;
    for (size_t r = 0; r < n_rounds; r++) {
      PcsOpenRound pr;
      pr.tree = (const MerkleTree*)mmcs[r];
      for (size_t m = 0; m < pr.tree->mats.size(); m++) {
        std::vector<EF> pts;
        for (u64 k = 0; k < n_points[mk]; k++) pts.push_back(ef_from(points + EXT_D * pk++));
        mk++;
        pr.points.push_back(std::move(pts));
      }
      rounds.push_back(std::move(pr));
    }
    std::vector<OpenedRound> opened;
    FriProof proof;
    pcs_open_rounds(prm, rounds, *(Challenger*)challenger, opened, proof);
    size_t k = 0;
    for (auto& orr : opened)
      for (auto& m : orr)
        for (auto& pt : m)
          for (auto& e : pt) {
            ef_to(e, opened_out + k);
            k += EXT_D;
          }
    std::vector<uint8_t> bytes = fri_to_bytes(proof);
    if (bytes.size() > fri_cap) return -(long)bytes.size();
    memcpy(fri_out, bytes.data(), bytes.size());
    return (long)bytes.size();
  } catch (const std::exception& e) {
    g_err = e.what();
    return -1;
  }
}
int mso_pcs_verify(const u64* params7, size_t n_rounds, const uint8_t* const* caps, const u64* cap_sizes, const u64* n_mats, const u64* log_n,
                   const u64* widths, const u64* n_points, const u64* points, const u64* opened, const uint8_t* fri, size_t fri_len,
                   void* challenger) {
  TRY Params prm = params_from(params7);
  std::vector<PcsVerifyRound> rounds;
  size_t mk = 0, pk = 0, ok = 0;
  for (size_t r = 0; r < n_rounds; r++) {
    PcsVerifyRound vr;
    vr.commit.resize(cap_sizes[r]);
    for (size_t i = 0; i < cap_sizes[r]; i++) memcpy(vr.commit[i].b, caps[r] + 32 * i, 32);
    for (u64 m = 0; m < n_mats[r]; m++) {
      vr.log_n.push_back((unsigned)log_n[mk]);
      std::vector<std::pair<EF, std::vector<EF>>> pts;
      for (u64 k = 0; k < n_points[mk]; k++) {
        std::vector<EF> vals;
        for (u64 c = 0; c < widths[mk]; c++) {
          vals.push_back(ef_from(opened + ok));
          ok += EXT_D;
        }
        pts.emplace_back(ef_from(points + EXT_D * pk++), std::move(vals));
      }
      mk++;
      vr.mats.push_back(std::move(pts));
    }
    rounds.push_back(std::move(vr));
  }
  FriProof proof;
  try {
    proof = fri_from_bytes(fri, fri_len);
  } catch (const std::exception&) {
    return 0;
  }
  return pcs_verify_rounds(prm, rounds, proof, *(Challenger*)challenger) ? 1 : 0;
  CATCH
}
void mso_challenger_observe_digests(void* c, const uint8_t* d, size_t n) {
  for (size_t i = 0; i < n; i++) {
    Digest x;
    memcpy(x.b, d + 32 * i, 32);
    ((Challenger*)c)->observe_digest(x);
  }
}
// config.initialise_challenger() for the seven parameters (src/types.rs:118-130; baby_bear_config.rs:72-86,108-114)
void* mso_challenger_for_params(const u64* params7) {
  System s;
  s.params = Params();
  s.params.log_blowup = params7[0], s.params.cap_height = params7[1], s.params.log_final_poly_len = params7[2];
  s.params.max_log_arity = params7[3], s.params.num_queries = params7[4], s.params.commit_pow_bits = params7[5];
  s.params.query_pow_bits = params7[6];
  return new Challenger(s.new_challenger());
}

// ---- system / witness / prove / verify
void* mso_system_create(const uint8_t* blob, size_t len) {
  try {
    return new System(system_from_blob(blob, len));
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void mso_system_free(void* s) { delete (System*)s; }
// returns number of cap digests written (0 if no preprocessed commitment)
int mso_system_preprocessed_commit(void* s, uint8_t* out) {
  System* sys = (System*)s;
  if (!sys->has_pre) return 0;
  for (size_t i = 0; i < sys->pre_commit.size(); i++) memcpy(out + 32 * i, sys->pre_commit[i].b, 32);
  return (int)sys->pre_commit.size();
}
// circuit info: [main_width, pre_width, pre_height, num_lookups, stage2_width, constraint_count,
//                max_constraint_degree, quotient_degree, args_width]
int mso_system_circuit_info(void* s, size_t ci, u64* out9) {
  TRY System* sys = (System*)s;
  if (ci >= sys->circuits.size()) throw std::runtime_error("circuit index out of range");
  const Circuit& c = sys->circuits[ci];
  size_t aw = 0;
  for (auto& l : c.lookups) aw += l.args.size();
  u64 v[9] = {c.main_width, c.pre_width,        c.pre_height,           c.num_lookups,        c.stage2_width,
              c.constraint_count, c.max_constraint_degree, c.quotient_degree(), aw};
  memcpy(out9, v, sizeof(v));
  return 0;
  CATCH
}

// lookup values of SystemWitness::from_stage_1 for one circuit (src/system.rs:275-328)
int mso_compute_lookup_values(void* s, size_t ci, const u64* trace, size_t height, u64* mult_out, u64* args_out) {
  TRY System* sys = (System*)s;
  std::vector<Mat> traces(sys->circuits.size());
  traces[ci] = mat_from(trace, height, sys->circuits[ci].main_width);
  // evaluate only circuit ci: give the others empty traces
  System& S = *sys;
  Witness w = witness_from_stage_1(S, std::move(traces));
  const LookupValues& lv = w.lookups[ci];
  if (!lv.mult.empty()) memcpy(mult_out, lv.mult.data(), lv.mult.size() * 8);
  if (!lv.args.empty()) memcpy(args_out, lv.args.data(), lv.args.size() * 8);
  return 0;
  CATCH
}

// traces[i]: row-major heights[i] x main_width_i (height 0 = inactive). times_out: 6 doubles or null.
// Returns proof length (bytes) or -1; if cap is too small returns the needed length negated minus 1.
long mso_prove(void* s, size_t n_claims, const u64* claim_offsets, const u64* claim_data, const u64* const* traces,
               const u64* heights, uint8_t* proof_out, size_t cap, double* times_out) {
  TRY System* sys = (System*)s;
  std::vector<Mat> tr;
  for (size_t i = 0; i < sys->circuits.size(); i++) tr.push_back(mat_from(traces[i], heights[i], sys->circuits[i].main_width));
  Witness w = witness_from_stage_1(*sys, std::move(tr));
  StageTimes st;
  Proof p = prove(*sys, unpack_claims(n_claims, claim_offsets, claim_data), std::move(w), &st);
  std::vector<uint8_t> bytes = proof_to_bytes(p);
  if (times_out) {
    double t[6] = {st.stage1_commit, st.lookup_construction, st.stage2_commit, st.quotient, st.fri_open, st.total};
    memcpy(times_out, t, sizeof(t));
  }
  if (bytes.size() > cap) return -(long)bytes.size() - 1;
  memcpy(proof_out, bytes.data(), bytes.size());
  return (long)bytes.size();
  CATCH
}

// 0 = accepted; VerifyError code otherwise; -1 = malformed bytes / exception
int mso_verify(void* s, size_t n_claims, const u64* claim_offsets, const u64* claim_data, const uint8_t* proof,
               size_t len) {
  TRY System* sys = (System*)s;
  Proof p = proof_from_bytes(proof, len);
  return (int)verify(*sys, unpack_claims(n_claims, claim_offsets, claim_data), p);
  CATCH
}

// ---- kernel-level pieces
// lookups of ONE circuit -> stage-2 trace (height x max(L,1) Ext2, row-major, c0,c1 interleaved) and acc_out
int mso_stage2_trace(size_t height, size_t num_lookups, const u64* mult, const u64* arg_offsets, const u64* args,
                     const u64* beta, const u64* gamma, const u64* acc_in, u64* trace_out, u64* acc_out) {
  TRY LookupValues lv;
  lv.height = height;
  lv.num_lookups = num_lookups;
  lv.arg_offsets.assign(arg_offsets, arg_offsets + num_lookups + 1);
  lv.mult.assign(mult, mult + height * num_lookups);
  lv.args.assign(args, args + height * lv.arg_offsets.back());
  std::vector<std::vector<EF>> tr;
  std::vector<EF> accs;
  std::vector<LookupValues> cs;
  cs.push_back(std::move(lv));
  stage_2_traces(cs, ef_from(beta), ef_from(gamma), ef_from(acc_in), tr, accs);
  for (size_t i = 0; i < tr[0].size(); i++) ef_to(tr[0][i], trace_out + EXT_D * i);
  ef_to(accs[0], acc_out);
  return 0;
  CATCH
}
int mso_claims_accumulator(size_t n_claims, const u64* claim_offsets, const u64* claim_data, const u64* beta,
                           const u64* gamma, u64* acc_out) {
  TRY EF a = claims_accumulator(unpack_claims(n_claims, claim_offsets, claim_data), ef_from(beta), ef_from(gamma));
  ef_to(a, acc_out);
  return 0;
  CATCH
}
// natural-order trace evaluations on the quotient domain (nq rows each) -> nq Ext2 quotient values
int mso_quotient_values(void* s, size_t ci, const u64* publics8, unsigned log_n, unsigned log_q, const u64* pre_q,
                        const u64* s1_q, const u64* s2_q, const u64* alpha, u64* out) {
  TRY System* sys = (System*)s;
  const Circuit& c = sys->circuits[ci];
  size_t N = size_t(1) << (log_n + log_q);
  Mat pre = c.pre_width ? mat_from(pre_q, N, c.pre_width) : Mat();
  Mat s1 = mat_from(s1_q, N, c.main_width), s2 = mat_from(s2_q, N, c.stage2_width);
  std::vector<EF> q = quotient_values(c, publics8, log_n, log_q, c.pre_width ? &pre : nullptr, s1, s2, ef_from(alpha));
  for (size_t i = 0; i < N; i++) ef_to(q[i], out + EXT_D * i);
  return 0;
  CATCH
}
int mso_selectors_on_coset(unsigned log_n, unsigned log_q, u64* is_first, u64* is_last, u64* is_trans, u64* inv_van) {
  TRY Selectors s = selectors_on_coset(log_n, log_q);
  size_t N = s.is_first.size();
  memcpy(is_first, s.is_first.data(), N * 8);
  memcpy(is_last, s.is_last.data(), N * 8);
  memcpy(is_trans, s.is_trans.data(), N * 8);
  memcpy(inv_van, s.inv_van.data(), N * 8);
  return 0;
  CATCH
}

}  // extern "C"
