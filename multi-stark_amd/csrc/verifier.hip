// System::verify_multiple_claims for GoldilocksBlake3Config (/root/reference/src/verifier.rs:208-532, shape checks
// :536-695) over the proof bytes ms_prove emits: the step after the hot path. Verification is light
// except for the claims: the transcript absorbs every claim (42 MB at the bench size) and the initial accumulator
// inverts one fingerprint per claim - those two run on the device with the prover's own kernels (BLAKE3 tree hash,
// batched inversion), the rest (transcript replay, Merkle paths, reduced openings, FRI fold chain, out-of-domain
// check of every circuit's constraints at zeta) is host code. The PCS part restates p3-fri 0.5.1
// TwoAdicFriPcs::verify / verify_fri / verify_query [upstream, from the published algorithm; parity unpinned like
// the prover's side - see DESIGN.md §2].
// Error codes follow the reference's VerificationError variants (src/verifier.rs:176-192).
#include <algorithm>
#include <cstring>
#include <map>
#include <stdexcept>

#include "host.h"

namespace msamd {

namespace {

enum : int { V_OK = 0, V_INVALID_OPENING = 2, V_INVALID_SHAPE = 3, V_INVALID_SYSTEM = 4, V_OOD_MISMATCH = 5, V_UNBALANCED = 6 };

struct Malformed {};  // thrown by the reader: truncated or oversized fields -> InvalidProofShape

struct Reader {
  const uint8_t* p;
  size_t n, pos = 0;
  void need(size_t k) const {
    if (k > n - pos) throw Malformed();
  }
  uint8_t u8() {
    need(1);
    return p[pos++];
  }
  u64 u64_() {
    need(8);
    u64 v;
    memcpy(&v, p + pos, 8);
    pos += 8;
    return v;
  }
  size_t count(size_t elem_bytes) {  // a length prefix whose payload must still fit
    u64 c = u64_();
    if (elem_bytes && c > (n - pos) / elem_bytes) throw Malformed();
    return (size_t)c;
  }
  u64 field() {
    u64 v = u64_();
    if (v >= GL_P) throw Malformed();
    return v;
  }
  E2 ext() {
    E2 e;
    e.c0 = field();
    e.c1 = field();
    return e;
  }
  Digest digest() {
    need(32);
    Digest d;
    memcpy(d.b, p + pos, 32);
    pos += 32;
    return d;
  }
  std::vector<Digest> cap() {
    size_t c = count(32);
    std::vector<Digest> v(c);
    for (auto& d : v) d = digest();
    return v;
  }
};

typedef std::vector<std::vector<std::vector<E2>>> OpenedRound;  // matrix -> point -> values
OpenedRound read_round(Reader& r) {
  OpenedRound out(r.count(8));
  for (auto& m : out) {
    m.resize(r.count(8));
    for (auto& pt : m) {
      pt.resize(r.count(16));
      for (auto& e : pt) e = r.ext();
    }
  }
  return out;
}

struct BatchOpening {
  std::vector<std::vector<u64>> rows;  // one opened row per matrix of the batch
  std::vector<Digest> path;
};
struct FriStep {
  unsigned log_arity = 1;
  std::vector<E2> siblings;  // the opened row without the queried position's own value: 2^log_arity - 1 values
  std::vector<Digest> path;
};
struct QueryProof {
  std::vector<BatchOpening> inputs;
  std::vector<FriStep> steps;
};
struct FriProofV {
  std::vector<std::vector<Digest>> commits;
  std::vector<u64> pow;
  std::vector<QueryProof> queries;
  std::vector<E2> final_poly;
  u64 query_pow = 0;
};
struct ProofV {
  std::vector<uint8_t> active;
  std::vector<Digest> s1, s2, q;
  std::vector<E2> accs;
  std::vector<uint8_t> log_degrees;
  FriProofV fri;
  OpenedRound q_opened, pre_opened, s1_opened, s2_opened;
  bool has_pre = false;
};

void parse_fri(Reader& r, FriProofV& f) {
  f.commits.resize(r.count(8));
  for (auto& c : f.commits) c = r.cap();
  f.pow.resize(r.count(8));
  for (auto& w : f.pow) w = r.u64_();
  f.queries.resize(r.count(8));
  for (auto& q : f.queries) {
    q.inputs.resize(r.count(8));
    for (auto& bo : q.inputs) {
      bo.rows.resize(r.count(8));
      for (auto& row : bo.rows) {
        row.resize(r.count(8));
        for (auto& v : row) v = r.field();
      }
      bo.path.resize(r.count(32));
      for (auto& d : bo.path) d = r.digest();
    }
    q.steps.resize(r.count(8));
    for (auto& st : q.steps) {
      st.log_arity = r.u8();
      if (st.log_arity < 1 || st.log_arity > FRI_MAX_LOG_ARITY) throw Malformed();
      st.siblings.resize(r.count(16));
      if (st.siblings.size() != (size_t(1) << st.log_arity) - 1) throw Malformed();
      for (auto& e : st.siblings) e = r.ext();
      st.path.resize(r.count(32));
      for (auto& d : st.path) d = r.digest();
    }
  }
  f.final_poly.resize(r.count(16));
  for (auto& e : f.final_poly) e = r.ext();
  f.query_pow = r.u64_();
}

ProofV parse(const uint8_t* bytes, size_t len) {
  Reader r{bytes, len};
  ProofV p;
  p.active.resize(r.count(1));
  for (auto& a : p.active) {
    a = r.u8();
    if (a > 1) throw Malformed();  // bincode decodes a bool from 0 or 1 only
  }
  p.s1 = r.cap();
  p.s2 = r.cap();
  p.q = r.cap();
  p.accs.resize(r.count(16));
  for (auto& a : p.accs) a = r.ext();
  p.log_degrees.resize(r.count(1));
  for (auto& l : p.log_degrees) l = r.u8();
  parse_fri(r, p.fri);
  p.q_opened = read_round(r);
  {
    const uint8_t tag = r.u8();  // Option tag
    if (tag > 1) throw Malformed();
    p.has_pre = tag != 0;
  }
  if (p.has_pre) p.pre_opened = read_round(r);
  p.s1_opened = read_round(r);
  p.s2_opened = read_round(r);
  if (r.pos != len) throw Malformed();
  return p;
}

Digest hash_elems(const std::vector<u64>& v) {
  Digest d;
  blake3_host(reinterpret_cast<const uint8_t*>(v.data()), v.size() * 8, d.b);  // canonical u64, little-endian host
  return d;
}
Digest compress2(const Digest& l, const Digest& r) {
  uint8_t buf[64];
  memcpy(buf, l.b, 32);
  memcpy(buf + 32, r.b, 32);
  Digest d;
  blake3_host(buf, 64, d.b);
  return d;
}
bool same(const Digest& a, const Digest& b) { return memcmp(a.b, b.b, 32) == 0; }

struct Dim {
  size_t w, h;
};
// MerkleTreeMmcs::verify_batch: rows of the tallest matrices form the leaf, shorter ones are injected on the way up
bool mmcs_verify_batch(const std::vector<Digest>& cap, const std::vector<Dim>& dims, size_t index, const BatchOpening& o) {
  if (dims.size() != o.rows.size() || dims.empty()) return false;
  std::vector<size_t> order(dims.size());
  for (size_t i = 0; i < dims.size(); i++) {
    order[i] = i;
    if (o.rows[i].size() != dims[i].w) return false;
    if (dims[i].h == 0 || (dims[i].h & (dims[i].h - 1))) return false;
  }
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return dims[a].h > dims[b].h; });
  size_t pos = 0, cur = dims[order[0]].h;
  const unsigned log_max = log2_strict(cur);
  auto hash_group = [&](size_t height) {
    std::vector<u64> buf;
    while (pos < order.size() && dims[order[pos]].h == height) {
      auto& v = o.rows[order[pos]];
      buf.insert(buf.end(), v.begin(), v.end());
      pos++;
    }
    return hash_elems(buf);
  };
  Digest root = hash_group(cur);
  const size_t capn = cap.size();
  if (capn == 0 || (capn & (capn - 1))) return false;
  const unsigned ch = log2_strict(capn);
  if (ch > log_max || o.path.size() != log_max - ch) return false;
  size_t idx = index;
  if (idx >= (size_t(1) << log_max)) return false;
  for (auto& sib : o.path) {
    root = (idx & 1) ? compress2(sib, root) : compress2(root, sib);
    idx >>= 1;
    cur >>= 1;
    if (pos < order.size() && dims[order[pos]].h == cur) root = compress2(root, hash_group(cur));
  }
  if (pos != order.size()) return false;
  return same(root, cap[idx]);
}

bool check_witness(Challenger& ch, unsigned bits, u64 w) {
  if (bits == 0) return true;  // DeterministicPow: nothing is observed at zero bits (src/types.rs:75-80)
  if (w >= GL_P) return false;
  ch.observe(w);
  return ch.sample_bits(bits) == 0;
}

struct RoundClaim {
  std::vector<Digest> commit;
  std::vector<unsigned> log_n;                                       // per matrix: log2 of the trace height
  std::vector<std::vector<std::pair<E2, const std::vector<E2>*>>> mats;  // per matrix: (point, claimed values)
};

bool pcs_verify(const Params& prm, const std::vector<RoundClaim>& rounds, const FriProofV& proof, Challenger& ch) {
  const unsigned lb = (unsigned)prm.log_blowup;
  for (auto& r : rounds)
    for (auto& m : r.mats)
      for (auto& pv : m)
        for (auto& y : *pv.second) ch.observe_ext(y);
  const E2 alpha = ch.sample_ext();
  const size_t nrounds = proof.commits.size();
  if (proof.pow.size() != nrounds) return false;
  // every query repeats the rounds' arities; the first one's place the tallest input, and each is checked below against
  // what the prover had to choose (p3-fri compute_log_arity_for_round) once the input heights are known
  std::vector<unsigned> arities(nrounds, 1);
  if (!proof.queries.empty()) {
    if (proof.queries[0].steps.size() != nrounds) return false;
    for (size_t i = 0; i < nrounds; i++) arities[i] = proof.queries[0].steps[i].log_arity;
  }
  unsigned log_gmax = (unsigned)(lb + prm.log_final_poly_len);
  for (unsigned a : arities) {
    if (a > prm.max_log_arity) return false;
    log_gmax += a;
  }
  if (log_gmax > GL_TWO_ADICITY) return false;  // no subgroup of that order: gl_two_adic_generator is defined up to 2^32
  std::vector<E2> betas;
  for (size_t i = 0; i < nrounds; i++) {
    ch.observe_cap(proof.commits[i]);
    if (!check_witness(ch, (unsigned)prm.commit_pow_bits, proof.pow[i])) return false;
    betas.push_back(ch.sample_ext());
  }
  if (proof.final_poly.size() != (size_t(1) << prm.log_final_poly_len)) return false;
  for (auto& c : proof.final_poly) ch.observe_ext(c);
  if (proof.queries.size() != prm.num_queries) return false;
  if (!check_witness(ch, (unsigned)prm.query_pow_bits, proof.query_pow)) return false;
  const unsigned log_final_height = (unsigned)(lb + prm.log_final_poly_len);
  for (auto& qp : proof.queries) {
    const size_t index = ch.sample_bits(log_gmax);
    if (qp.inputs.size() != rounds.size()) return false;
    std::map<unsigned, std::pair<E2, E2>> ro;  // log height -> (running alpha power, reduced opening)
    for (size_t ri = 0; ri < rounds.size(); ri++) {
      const RoundClaim& r = rounds[ri];
      const BatchOpening& bo = qp.inputs[ri];
      if (bo.rows.size() != r.mats.size()) return false;
      std::vector<Dim> dims;
      unsigned log_bmax = 0;
      for (size_t mi = 0; mi < r.mats.size(); mi++) {
        dims.push_back(Dim{bo.rows[mi].size(), size_t(1) << (r.log_n[mi] + lb)});
        log_bmax = std::max(log_bmax, r.log_n[mi] + lb);
      }
      if (log_bmax > log_gmax) return false;
      if (!mmcs_verify_batch(r.commit, dims, index >> (log_gmax - log_bmax), bo)) return false;
      for (size_t mi = 0; mi < r.mats.size(); mi++) {
        const unsigned lh = r.log_n[mi] + lb;
        const size_t rev = bitrev64(index >> (log_gmax - lh), lh);
        const u64 x = gl_mul(GL_GEN, gl_pow(gl_two_adic_generator(lh), rev));
        auto it = ro.find(lh);
        if (it == ro.end()) it = ro.emplace(lh, std::make_pair(e2(1), e2(0))).first;
        for (auto& pv : r.mats[mi]) {
          if (pv.second->size() != bo.rows[mi].size()) return false;
          const E2 quot = e2_inv(e2_sub(pv.first, e2(x)));
          for (size_t c = 0; c < pv.second->size(); c++) {
            const E2 diff = e2_sub((*pv.second)[c], e2(bo.rows[mi][c]));
            it->second.second = e2_add(it->second.second, e2_mul(e2_mul(it->second.first, diff), quot));
            it->second.first = e2_mul(it->second.first, alpha);
          }
        }
      }
    }
    // a height-1 trace gives a constant polynomial: its reduced opening must vanish
    auto low = ro.find(lb);
    if (low != ro.end() && log_final_height >= lb && lb < log_gmax) {
      if (!e2_is_zero(low->second.second)) return false;
      ro.erase(low);
    }
    if (qp.steps.size() != nrounds) return false;
    auto it = ro.rbegin();
    if (it == ro.rend() || it->first != log_gmax) return false;
    E2 folded = it->second.second;
    ++it;
    size_t idx = index;
    unsigned log_height = log_gmax;
    for (size_t i = 0; i < nrounds; i++) {
      const FriStep& st = qp.steps[i];
      const unsigned la = st.log_arity;
      if (la != arities[i] || log_height <= log_final_height) return false;
      {  // the schedule: as far as max_log_arity allows without stepping over the next input or below the final height
        unsigned want = std::min<unsigned>((unsigned)prm.max_log_arity, log_height - log_final_height);
        if (it != ro.rend()) want = std::min(want, log_height - it->first);
        if (la != want) return false;
      }
      const unsigned log_folded_height = log_height - la;
      if (la == 1) {
        const size_t sib = idx ^ 1, pair = idx >> 1;
        E2 evals[2];
        evals[idx % 2] = folded;
        evals[sib % 2] = st.siblings[0];
        BatchOpening bo;
        bo.rows.push_back({evals[0].c0, evals[0].c1, evals[1].c0, evals[1].c1});  // ExtensionMmcs: flattened row
        bo.path = st.path;
        if (!mmcs_verify_batch(proof.commits[i], {Dim{4, size_t(1) << log_folded_height}}, pair, bo)) return false;
        idx = pair;
        // fold_row: the line through (x0, e0), (-x0, e1) evaluated at beta; x0 = w^bitrev(idx) on the subgroup
        const u64 x0 = gl_pow(gl_two_adic_generator(log_folded_height + 1), bitrev64(idx, log_folded_height));
        const u64 x1 = gl_neg(x0);
        const E2 slope = e2_mul_base(e2_sub(evals[1], evals[0]), gl_inv(gl_sub(x1, x0)));
        folded = e2_add(evals[0], e2_mul(e2_sub(betas[i], e2(x0)), slope));
      } else {
        // a row of 2^la values: position j holds the value at x w^bitrev(j), w of order 2^la, x = w_{2^log_height}^bitrev(row);
        // fold_row is the polynomial of degree < 2^la through them at beta - barycentric form over the coset x <w>:
        // p(beta) = (beta^m - x^m) / (m x^m) * sum_j e_j h_j / (beta - h_j)
        const size_t m = size_t(1) << la, own = idx & (m - 1), row = idx >> la;
        std::vector<E2> evals(m);
        for (size_t j = 0, k = 0; j < m; j++) evals[j] = j == own ? folded : st.siblings[k++];
        BatchOpening bo;
        bo.rows.emplace_back();
        for (auto& e : evals) bo.rows[0].push_back(e.c0), bo.rows[0].push_back(e.c1);
        bo.path = st.path;
        if (!mmcs_verify_batch(proof.commits[i], {Dim{2 * m, size_t(1) << log_folded_height}}, row, bo)) return false;
        idx = row;
        const u64 x = gl_pow(gl_two_adic_generator(log_height), bitrev64(row, log_folded_height));
        const u64 wm = gl_two_adic_generator(la);
        const E2 beta = betas[i];
        E2 sum = e2(0);
        bool hit = false;
        for (size_t j = 0; j < m && !hit; j++) {
          const u64 h = gl_mul(x, gl_pow(wm, bitrev64(j, la)));
          const E2 d = e2_sub(beta, e2(h));
          if (e2_is_zero(d)) {  // beta is one of the row's points
            folded = evals[j];
            hit = true;
          } else {
            sum = e2_add(sum, e2_mul(e2_mul_base(evals[j], h), e2_inv(d)));
          }
        }
        if (!hit) {
          const u64 xm = gl_pow(x, m);
          const E2 z = e2_sub(e2_exp_pow2(beta, la), e2(xm));
          folded = e2_mul(e2_mul_base(z, gl_inv(gl_mul(xm, (u64)m))), sum);
        }
      }
      log_height = log_folded_height;
      if (it != ro.rend() && it->first == log_folded_height) {
        // roll-in factor: the next power of beta after the 2^la the fold used (beta^2 for a binary round)
        folded = e2_add(folded, e2_mul(e2_exp_pow2(betas[i], la), it->second.second));
        ++it;
      }
    }
    if (it != ro.rend()) return false;
    const u64 x = gl_pow(gl_two_adic_generator(log_gmax), bitrev64(idx, log_gmax));
    E2 eval = e2(0);
    for (size_t k = proof.final_poly.size(); k-- > 0;) eval = e2_add(e2_mul_base(eval, x), proof.final_poly[k]);
    if (!(eval.c0 == folded.c0 && eval.c1 == folded.c1)) return false;
  }
  return true;
}

// the (c0 + c1 X)(d0 + d1 X) of src/lookup.rs:123-128 over extension-field coordinates
void mul2e(E2 a0, E2 a1, E2 b0, E2 b1, E2& c0, E2& c1) {
  const E2 v0 = e2_mul(a0, b0), v1 = e2_mul(a1, b1);
  c1 = e2_sub(e2_sub(e2_mul(e2_add(a0, a1), e2_add(b0, b1)), v0), v1);
  c0 = e2_add(v0, e2_mul_base(v1, GL_EXT_W));
}

}  // namespace

// Pcs::verify on its own (examples/pcs_example.rs:111-121): rounds given as commitments, domain sizes, points and the
// claimed values (flat: round -> matrix -> point -> column)
bool pcs_verify_standalone(const Params& prm, const std::vector<std::vector<Digest>>& commits, const std::vector<std::vector<unsigned>>& log_n,
                           const std::vector<std::vector<size_t>>& widths, const std::vector<std::vector<std::vector<E2>>>& points,
                           const std::vector<E2>& opened_flat, const uint8_t* fri, size_t fri_len, Challenger& ch) {
  if (prm.max_log_arity < 1 || prm.max_log_arity > FRI_MAX_LOG_ARITY || prm.log_blowup < 1 || prm.log_blowup > 8) return false;
  FriProofV proof;
  try {
    Reader r{fri, fri_len};
    parse_fri(r, proof);
    if (r.pos != fri_len) return false;
  } catch (const Malformed&) {
    return false;
  }
  std::vector<std::vector<std::vector<std::vector<E2>>>> vals(commits.size());  // round -> matrix -> point -> values
  size_t k = 0;
  for (size_t r = 0; r < commits.size(); r++) {
    vals[r].resize(points[r].size());
    for (size_t m = 0; m < points[r].size(); m++)
      for (size_t p = 0; p < points[r][m].size(); p++) {
        if (k + widths[r][m] > opened_flat.size()) return false;
        vals[r][m].emplace_back(opened_flat.begin() + k, opened_flat.begin() + k + widths[r][m]);
        k += widths[r][m];
      }
  }
  if (k != opened_flat.size()) return false;
  std::vector<RoundClaim> rounds(commits.size());
  for (size_t r = 0; r < commits.size(); r++) {
    rounds[r].commit = commits[r];
    rounds[r].log_n = log_n[r];
    for (size_t m = 0; m < points[r].size(); m++) {
      std::vector<std::pair<E2, const std::vector<E2>*>> pv;
      for (size_t p = 0; p < points[r][m].size(); p++) pv.emplace_back(points[r][m][p], &vals[r][m][p]);
      rounds[r].mats.push_back(std::move(pv));
    }
  }
  return pcs_verify(prm, rounds, proof, ch);
}

int verify(HSystem& sys, size_t n_claims, const u64* claim_offsets, const u64* claim_data, const uint8_t* proof_bytes, size_t proof_len) {
  Ctx& ctx = *sys.ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  const Params& prm = sys.params;
  const size_t C = sys.circuits.size();
  if (C == 0) return V_INVALID_SYSTEM;
  ProofV proof;
  try {
    proof = parse(proof_bytes, proof_len);
  } catch (const Malformed&) {
    return V_INVALID_SHAPE;
  }
  // ---- verify_shape (src/verifier.rs:536-695)
  if (proof.active.size() != C) return V_INVALID_SHAPE;
  std::vector<size_t> aidx;
  std::vector<int> apos(C, -1);
  for (size_t i = 0; i < C; i++)
    if (proof.active[i]) {
      apos[i] = (int)aidx.size();
      aidx.push_back(i);
    }
  const size_t na = aidx.size();
  if (na == 0 || proof.log_degrees.size() != na) return V_INVALID_SHAPE;
  size_t num_pre = 0;
  for (int pi : sys.pre_indices) num_pre += pi >= 0;
  if (sys.has_pre != (num_pre != 0)) return V_INVALID_SYSTEM;
  if ((proof.has_pre ? proof.pre_opened.size() : 0) != num_pre) return V_INVALID_SHAPE;
  for (size_t ci = 0; ci < C; ci++)
    if (sys.pre_indices[ci] >= 0 && !proof.active[ci] && proof.pre_opened[sys.pre_indices[ci]].size() != 0) return V_INVALID_SHAPE;
  if (proof.s1_opened.size() != na || proof.s2_opened.size() != na || proof.q_opened.size() != na) return V_INVALID_SHAPE;
  std::vector<size_t> qdeg;
  for (size_t pos = 0; pos < na; pos++) {
    const size_t ci = aidx[pos];
    const HCircuit& c = sys.circuits[ci];
    const int slot = sys.pre_indices[ci];
    if (proof.s1_opened[pos].size() != 2 || proof.s2_opened[pos].size() != 2) return V_INVALID_SHAPE;
    if (slot >= 0 && proof.pre_opened[slot].size() != 2) return V_INVALID_SHAPE;
    for (int j = 0; j < 2; j++) {
      if (slot >= 0 && proof.pre_opened[slot][j].size() != c.pre_width) return V_INVALID_SHAPE;
      if (proof.s1_opened[pos][j].size() != c.main_width) return V_INVALID_SHAPE;
      if (proof.s2_opened[pos][j].size() != c.stage2_width) return V_INVALID_SHAPE;
    }
    const size_t qd = c.quotient_degree();
    if (proof.log_degrees[pos] + log2_strict(qd) > 32 - prm.log_blowup) return V_INVALID_SHAPE;  // src/types.rs:131
    if (c.pre_width && (size_t(1) << proof.log_degrees[pos]) != c.pre_height) return V_INVALID_SHAPE;
    qdeg.push_back(qd);
    if (proof.q_opened[pos].size() != 1 || proof.q_opened[pos][0].size() != qd * 2) return V_INVALID_SHAPE;
  }
  if (proof.accs.size() != na) return V_INVALID_SHAPE;
  if (!e2_is_zero(proof.accs.back())) return V_UNBALANCED;  // src/verifier.rs:242-246

  // ---- transcript replay (src/verifier.rs:255-326); the claims go through the device like in the prover
  for (size_t i = 0; i < n_claims; i++)
    if (claim_offsets[i + 1] < claim_offsets[i]) return V_INVALID_SHAPE;
  const size_t claim_elems = n_claims ? (size_t)claim_offsets[n_claims] : 0;
  for (size_t i = 0; i < claim_elems; i++)
    if (claim_data[i] >= GL_P) return V_INVALID_SHAPE;
  Challenger ch(sys.seed);
  ch.observe((u64)C);
  for (auto& c : sys.circuits) {
    ch.observe((u64)c.constraint_count);
    ch.observe((u64)c.max_constraint_degree);
    ch.observe((u64)c.pre_height);
    ch.observe((u64)c.pre_width);
    ch.observe((u64)c.main_width);
    ch.observe((u64)c.stage2_width);
  }
  for (auto a : proof.active) ch.observe(a ? 1 : 0);
  if (sys.has_pre) ch.observe_cap(sys.pre_commit);
  ch.observe_cap(proof.s1);
  for (auto ld : proof.log_degrees) ch.observe((u64)ld);
  const size_t claim_words = 1 + n_claims + claim_elems;
  const bool device_claims = claim_words > 8192;
  DBuf<u64> d_offs, d_data;
  if (device_claims) {
    d_offs = DBuf<u64>(ctx, n_claims + 1);
    d_data = DBuf<u64>(ctx, std::max<size_t>(claim_elems, 1));
    ctx.h2d(d_offs.p, claim_offsets, (n_claims + 1) * 8);
    if (claim_elems) ctx.h2d(d_data.p, claim_data, claim_elems * 8);
    DBuf<uint8_t> d_prefix(ctx, ch.input.size());
    ctx.h2d(d_prefix.p, ch.input.data(), ch.input.size());
    DBuf<u64> d_words(ctx, claim_words);
    claims_transcript_words(ctx, d_data.p, d_offs.p, n_claims, claim_elems, d_words.p);
    ch.flush_with(blake3_device(ctx, d_prefix.p, ch.input.size(), d_words.p, claim_words));
  } else {
    ch.observe((u64)n_claims);
    for (size_t i = 0; i < n_claims; i++) {
      ch.observe(claim_offsets[i + 1] - claim_offsets[i]);
      for (u64 k = claim_offsets[i]; k < claim_offsets[i + 1]; k++) ch.observe(claim_data[k]);
    }
  }
  const E2 beta = ch.sample_ext();
  ch.observe_ext(beta);
  const E2 gamma = ch.sample_ext();
  ch.observe_ext(gamma);
  ch.observe_cap(proof.s2);
  for (auto& a : proof.accs) ch.observe_ext(a);
  E2 acc = e2(0);
  if (device_claims) {
    acc = claims_accumulator(ctx, d_data.p, d_offs.p, n_claims, beta, gamma);
  } else {
    for (size_t i = 0; i < n_claims; i++) {
      E2 f = e2(0);
      for (u64 k = claim_offsets[i + 1]; k-- > claim_offsets[i];) f = e2_add(e2_mul(f, gamma), e2(claim_data[k]));
      const E2 m = e2_add(beta, f);
      if (e2_is_zero(m)) return V_INVALID_SHAPE;  // the reference would divide by zero here
      acc = e2_add(acc, e2_inv(m));
    }
  }
  const E2 alpha = ch.sample_ext();
  ch.observe_cap(proof.q);
  const E2 zeta = ch.sample_ext();

  std::vector<RoundClaim> rounds(3);
  rounds[0].commit = proof.s1;
  rounds[1].commit = proof.s2;
  rounds[2].commit = proof.q;
  for (size_t pos = 0; pos < na; pos++) {
    const unsigned ld = proof.log_degrees[pos];
    const E2 zn = e2_mul_base(zeta, gl_two_adic_generator(ld));
    rounds[0].log_n.push_back(ld);
    rounds[0].mats.push_back({{zeta, &proof.s1_opened[pos][0]}, {zn, &proof.s1_opened[pos][1]}});
    rounds[1].log_n.push_back(ld);
    rounds[1].mats.push_back({{zeta, &proof.s2_opened[pos][0]}, {zn, &proof.s2_opened[pos][1]}});
    rounds[2].log_n.push_back(ld);
    rounds[2].mats.push_back({{zeta, &proof.q_opened[pos][0]}});
  }
  if (sys.has_pre) {
    RoundClaim r0;
    r0.commit = sys.pre_commit;
    for (size_t ci = 0; ci < C; ci++) {
      const int slot = sys.pre_indices[ci];
      if (slot < 0) continue;
      if (apos[ci] >= 0) {
        const unsigned ld = proof.log_degrees[apos[ci]];
        const E2 zn = e2_mul_base(zeta, gl_two_adic_generator(ld));
        r0.log_n.push_back(ld);
        r0.mats.push_back({{zeta, &proof.pre_opened[slot][0]}, {zn, &proof.pre_opened[slot][1]}});
      } else {
        r0.log_n.push_back(log2_strict(sys.circuits[ci].pre_height));
        r0.mats.push_back({});
      }
    }
    rounds.push_back(std::move(r0));
  }
  if (!pcs_verify(prm, rounds, proof.fri, ch)) return V_INVALID_OPENING;

  // ---- out-of-domain check per circuit (src/verifier.rs:419-530): the constraints at zeta against the quotient
  for (size_t pos = 0; pos < na; pos++) {
    const size_t ci = aidx[pos];
    const HCircuit& c = sys.circuits[ci];
    const unsigned ld = proof.log_degrees[pos];
    const E2 next_acc = proof.accs[pos];
    const u64 g_inv = gl_inv(gl_two_adic_generator(ld));
    const E2 zh = e2_sub(e2_exp_pow2(zeta, ld), e2(1));  // selectors_at_point
    if (e2_is_zero(zh) || e2_is_zero(e2_sub(zeta, e2(1))) || e2_is_zero(e2_sub(zeta, e2(g_inv)))) return V_OOD_MISMATCH;
    const E2 is_first = e2_mul(zh, e2_inv(e2_sub(zeta, e2(1))));
    const E2 is_last = e2_mul(zh, e2_inv(e2_sub(zeta, e2(g_inv))));
    const E2 is_trans = e2_sub(zeta, e2(g_inv));
    const E2 inv_van = e2_inv(zh);
    const u64 inj_norm = gl_inv(gl_mul((u64(1) << ld) % GL_P, gl_two_adic_generator(ld)));
    const E2 four[4] = {beta, gamma, acc, next_acc};
    E2 publics[8];
    for (int k = 0; k < 4; k++) {
      publics[2 * k] = e2(four[k].c0);
      publics[2 * k + 1] = e2(four[k].c1);
    }
    const int slot = sys.pre_indices[ci];
    const std::vector<E2>* rows[3][2] = {{slot >= 0 ? &proof.pre_opened[slot][0] : nullptr, slot >= 0 ? &proof.pre_opened[slot][1] : nullptr},
                                         {&proof.s1_opened[pos][0], &proof.s1_opened[pos][1]},
                                         {&proof.s2_opened[pos][0], &proof.s2_opened[pos][1]}};
    std::vector<E2> buf(c.nodes.size());
    for (size_t i = 0; i < c.nodes.size(); i++) {  // ConstraintGraph::sweep_range over the extension field
      const PNode& n = c.nodes[i];
      E2 v;
      switch (n.kind) {
        case OP_CONST: v = e2(n.a); break;
        case OP_VAR: {
          if (n.source > 2 || n.offset > 1) return V_INVALID_SYSTEM;
          const std::vector<E2>* row = rows[n.source][n.offset];
          if (!row || n.a >= row->size()) return V_INVALID_SYSTEM;
          v = (*row)[n.a];
          break;
        }
        case OP_PUBLIC:
          if (n.a >= 8) return V_INVALID_SYSTEM;
          v = publics[n.a];
          break;
        case OP_IS_FIRST: v = is_first; break;
        case OP_IS_LAST: v = is_last; break;
        case OP_IS_TRANS: v = is_trans; break;
        case OP_ADD: v = e2_add(buf[n.a], buf[n.b]); break;
        case OP_SUB: v = e2_sub(buf[n.a], buf[n.b]); break;
        case OP_MUL: v = e2_mul(buf[n.a], buf[n.b]); break;
        default: v = e2_neg(buf[n.a]); break;
      }
      buf[i] = v;
    }
    std::vector<E2> cv;
    for (auto z : c.zeros) cv.push_back(buf[z]);
    // logup_constraint_values (src/lookup.rs:152-208)
    const std::vector<E2>&s2 = proof.s2_opened[pos][0], &s2n = proof.s2_opened[pos][1];
    const E2 ds0 = e2_mul_base(e2_sub(publics[6], publics[4]), inj_norm), ds1 = e2_mul_base(e2_sub(publics[7], publics[5]), inj_norm);
    const E2 inj0 = e2_mul(is_last, ds0), inj1 = e2_mul(is_last, ds1);
    if (c.lookups.empty()) {
      cv.push_back(e2_add(e2_sub(s2n[0], s2[0]), inj0));
      cv.push_back(e2_add(e2_sub(s2n[1], s2[1]), inj1));
    } else {
      const size_t last = c.lookups.size() - 1;
      for (size_t j = 0; j < c.lookups.size(); j++) {
        const auto& l = c.lookups[j];
        const E2 src0 = s2[2 * j], src1 = s2[2 * j + 1];
        const E2 tgt0 = j < last ? s2[2 * j + 2] : e2_add(s2n[0], inj0), tgt1 = j < last ? s2[2 * j + 3] : e2_add(s2n[1], inj1);
        E2 f0 = e2(0), f1 = e2(0);
        for (size_t k = l.second.size(); k-- > 0;) {
          E2 g0, g1;
          mul2e(f0, f1, publics[2], publics[3], g0, g1);
          f0 = e2_add(g0, buf[l.second[k]]);
          f1 = g1;
        }
        E2 c0, c1;
        mul2e(e2_add(f0, publics[0]), e2_add(f1, publics[1]), e2_sub(tgt0, src0), e2_sub(tgt1, src1), c0, c1);
        cv.push_back(e2_sub(c0, buf[l.first]));
        cv.push_back(c1);
      }
    }
    if (cv.size() != c.constraint_count) return V_INVALID_SYSTEM;
    E2 comp = e2(0);
    for (auto& x : cv) comp = e2_add(e2_mul(comp, alpha), x);
    // Q(zeta) = sum_i zeta^{i n} c_i(zeta), each chunk given by its two base-field coordinate polynomials
    const std::vector<E2>& qrow = proof.q_opened[pos][0];
    const E2 zpn = e2_exp_pow2(zeta, ld);
    E2 zp = e2(1), quot = e2(0);
    for (size_t i = 0; i < qdeg[pos]; i++) {
      const E2 chunk = e2_add(qrow[2 * i], e2_mul(qrow[2 * i + 1], e2(0, 1)));
      quot = e2_add(quot, e2_mul(zp, chunk));
      zp = e2_mul(zp, zpn);
    }
    const E2 lhs = e2_mul(comp, inv_van);
    if (!(lhs.c0 == quot.c0 && lhs.c1 == quot.c1)) return V_OOD_MISMATCH;
    acc = next_acc;
  }
  return V_OK;
}

}  // namespace msamd
