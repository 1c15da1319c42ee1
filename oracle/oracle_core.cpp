// ORACLE — test infrastructure only (see ms_oracle.hpp header). Hashing, DFT, Merkle MMCS, challenger,
// system-blob parsing and proof (de)serialisation.
#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>

#include "blake3_ref.hpp"
#include "ms_oracle.hpp"

namespace mso {

// ------------------------------------------------------------------ hashing
#ifndef MSO_BABYBEAR
Digest hash_bytes(const uint8_t* p, size_t n) {
  Digest d;
  blake3_hash(p, n, d.b);
  return d;
}

// p3 SerializingHasher<Blake3>::hash_iter: every field element -> 8 little-endian bytes of its canonical
// u64, concatenated, plain BLAKE3. [UPSTREAM-RECALL p3-symmetric 0.5.1]
Digest hash_elems(const u64* e, size_t n) {
  std::vector<uint8_t> buf(n * 8);
  for (size_t i = 0; i < n; i++)
    for (int k = 0; k < 8; k++) buf[8 * i + k] = (uint8_t)(e[i] >> (8 * k));
  return hash_bytes(buf.data(), buf.size());
}

// CompressionFunctionFromHasher<Blake3,2,32>::compress = BLAKE3(left || right) (src/types.rs:199)
Digest compress2(const Digest& l, const Digest& r) {
  uint8_t buf[64];
  memcpy(buf, l.b, 32);
  memcpy(buf + 32, r.b, 32);
  return hash_bytes(buf, 64);
}
#else
Poseidon2Constants& poseidon2_constants() {
  static Poseidon2Constants k = {};
  return k;
}
void poseidon2_permute(u64* s) { bb_poseidon2_permute(poseidon2_constants(), s); }

// BLAKE3 is not part of this configuration; kept so the primitive tests of the C surface still link
Digest hash_bytes(const uint8_t* p, size_t n) {
  Digest d;
  blake3_hash(p, n, d.b);
  return d;
}

// PaddingFreeSponge<Perm, 16, 8, 8>::hash_iter (baby_bear_config.rs:30): overwrite the first 8 state words with the
// next 8 inputs and permute; a partial last block is permuted too (the rest of the state keeps its old words), an
// input that ends on a block boundary - the empty input included - gets no extra permutation.
// [UPSTREAM-RECALL p3-symmetric 0.5.1 sponge.rs]
Digest hash_elems(const u64* e, size_t n) {
  u64 st[16] = {0};
  size_t i = 0;
  while (i < n) {
    size_t k = std::min<size_t>(8, n - i);
    for (size_t j = 0; j < k; j++) st[j] = e[i + j];
    poseidon2_permute(st);
    i += k;
  }
  Digest d;
  for (int j = 0; j < 8; j++) digest_set(d, j, st[j]);
  return d;
}

// TruncatedPermutation<Perm, 2, 8, 16>::compress (baby_bear_config.rs:31): permute(left || right)[..8]
Digest compress2(const Digest& l, const Digest& r) {
  u64 st[16];
  for (int j = 0; j < 8; j++) st[j] = digest_elem(l, j), st[8 + j] = digest_elem(r, j);
  poseidon2_permute(st);
  Digest d;
  for (int j = 0; j < 8; j++) digest_set(d, j, st[j]);
  return d;
}
#endif

// ------------------------------------------------------------------ DFT
namespace {
struct Twiddles {
  std::vector<u64> fwd, inv;  // w^i, w^-i for i < n/2
};
std::mutex g_tw_mu;
std::map<unsigned, Twiddles> g_tw;

const Twiddles& twiddles(unsigned logn) {
  std::lock_guard<std::mutex> lk(g_tw_mu);
  auto it = g_tw.find(logn);
  if (it != g_tw.end()) return it->second;
  Twiddles t;
  size_t half = logn ? (size_t(1) << (logn - 1)) : 0;
  t.fwd.resize(half);
  t.inv.resize(half);
  u64 w = f_two_adic_generator(logn), wi = f_inv(w);
  u64 a = 1, b = 1;
  for (size_t i = 0; i < half; i++) {
    t.fwd[i] = a;
    t.inv[i] = b;
    a = f_mul(a, w);
    b = f_mul(b, wi);
  }
  return g_tw.emplace(logn, std::move(t)).first->second;
}

// in-place radix-2 decimation-in-time transform of one column, natural order in and out
void ntt_column(u64* a, unsigned logn, const std::vector<u64>& tw) {
  size_t n = size_t(1) << logn;
  const bool par = n >= (size_t(1) << 14);
#pragma omp parallel for if (par) schedule(static)
  for (size_t i = 0; i < n; i++) {
    size_t j = bitrev(i, logn);
    if (i < j) std::swap(a[i], a[j]);
  }
  for (unsigned s = 1; s <= logn; s++) {
    size_t m = size_t(1) << s, half = m >> 1, step = n >> s;
#pragma omp parallel for if (par) schedule(static)
    for (size_t b = 0; b < n / 2; b++) {
      size_t k = (b / half) * m, j = b % half;
      u64 w = tw[j * step];
      u64 u = a[k + j], t = f_mul(w, a[k + j + half]);
      a[k + j] = f_add(u, t);
      a[k + j + half] = f_sub(u, t);
    }
  }
}

// column-major scratch helpers
std::vector<u64> to_cols(const Mat& m) {
  std::vector<u64> c(m.h * m.w);
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < m.h; r++)
    for (size_t j = 0; j < m.w; j++) c[j * m.h + r] = m.v[r * m.w + j];
  return c;
}
}  // namespace

Mat dft_batch(const Mat& m) {
  if (m.h <= 1) return m;
  unsigned logn = log2_strict(m.h);
  if ((size_t(1) << logn) != m.h) throw std::runtime_error("dft_batch: height not a power of two");
  const Twiddles& tw = twiddles(logn);
  std::vector<u64> c = to_cols(m);
  for (size_t j = 0; j < m.w; j++) ntt_column(&c[j * m.h], logn, tw.fwd);
  Mat o(m.h, m.w);
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < m.h; r++)
    for (size_t j = 0; j < m.w; j++) o.v[r * m.w + j] = c[j * m.h + r];
  return o;
}

Mat idft_batch(const Mat& m) {
  if (m.h <= 1) return m;
  unsigned logn = log2_strict(m.h);
  if ((size_t(1) << logn) != m.h) throw std::runtime_error("idft_batch: height not a power of two");
  const Twiddles& tw = twiddles(logn);
  std::vector<u64> c = to_cols(m);
  for (size_t j = 0; j < m.w; j++) ntt_column(&c[j * m.h], logn, tw.inv);
  u64 ninv = f_inv((u64)m.h);
  Mat o(m.h, m.w);
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < m.h; r++)
    for (size_t j = 0; j < m.w; j++) o.v[r * m.w + j] = f_mul(c[j * m.h + r], ninv);
  return o;
}

Mat bit_reverse_rows(const Mat& m) {
  unsigned logn = log2_strict(m.h);
  Mat o(m.h, m.w);
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < m.h; r++)
    memcpy(&o.v[bitrev(r, logn) * m.w], &m.v[r * m.w], m.w * sizeof(u64));
  return o;
}

// p3 TwoAdicSubgroupDft::coset_lde_batch(evals, added_bits, shift).bit_reverse_rows():
// iDFT -> coefficient j times shift^j -> zero-pad to n*2^added_bits -> DFT. [UPSTREAM-RECALL p3-dft;
// the result layout is pinned in-tree by src/prover.rs:975-999]
Mat coset_lde_bitrev(const Mat& evals, unsigned log_blowup, u64 shift) {
  size_t n = evals.h, N = n << log_blowup, w = evals.w;
  unsigned logn = log2_strict(n), logN = logn + log_blowup;
  if ((size_t(1) << logn) != n) throw std::runtime_error("coset_lde: height not a power of two");
  std::vector<u64> c(N * w, 0);  // column-major, zero padded
  {
    std::vector<u64> small = to_cols(evals);
    const Twiddles& tw = twiddles(logn);
    u64 ninv = f_inv((u64)n);
    // shift^j / n, per row
    std::vector<u64> sc(n);
    u64 s = ninv;
    for (size_t j = 0; j < n; j++) {
      sc[j] = s;
      s = f_mul(s, shift);
    }
    for (size_t j = 0; j < w; j++) {
      if (n > 1) ntt_column(&small[j * n], logn, tw.inv);
      u64* dst = &c[j * N];
      const u64* src = &small[j * n];
#pragma omp parallel for schedule(static)
      for (size_t r = 0; r < n; r++) dst[r] = f_mul(src[r], sc[r]);
    }
  }
  const Twiddles& twN = twiddles(logN);
  for (size_t j = 0; j < w; j++)
    if (N > 1) ntt_column(&c[j * N], logN, twN.fwd);
  Mat o(N, w);
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < N; r++) {
    size_t src = bitrev(r, logN);
    for (size_t j = 0; j < w; j++) o.v[r * w + j] = c[j * N + src];
  }
  return o;
}

// src/prover.rs:631-679
Mat shifted_quotient_slices(const Mat& quotient_evals, u64 domain_shift, size_t quotient_degree) {
  if (domain_shift != F_GENERATOR) throw std::runtime_error("quotient domain shift must equal the LDE shift");
  size_t ext_degree = quotient_evals.w, big = quotient_evals.h;
  unsigned log_big = log2_strict(big);
  size_t n = big / quotient_degree, width = quotient_degree * ext_degree;
  Mat storage = bit_reverse_rows(dft_batch(quotient_evals));  // natural k lives at row rev(k)
  u64 n_inv = f_inv((u64)big);
  u64 weight_step = f_inv(f_pow(F_GENERATOR, (u64)n));
  std::vector<u64> weights(quotient_degree);
  u64 wgt = 1;
  for (size_t k = 0; k < quotient_degree; k++) {
    weights[k] = f_mul(wgt, n_inv);
    wgt = f_mul(wgt, weight_step);
  }
  Mat out(n, width);
#pragma omp parallel for schedule(static)
  for (size_t row = 0; row < n; row++)
    for (size_t chunk = 0; chunk < quotient_degree; chunk++) {
      size_t j = chunk * n + row;
      size_t src = bitrev((big - j) & (big - 1), log_big);
      for (size_t c = 0; c < ext_degree; c++)
        out.v[row * width + chunk * ext_degree + c] = f_mul(storage.v[src * ext_degree + c], weights[chunk]);
    }
  return out;
}

// src/prover.rs:709-717
Mat lde_from_shifted_coefficients(const Mat& coeffs, unsigned log_blowup) {
  Mat padded(coeffs.h << log_blowup, coeffs.w);
  std::copy(coeffs.v.begin(), coeffs.v.end(), padded.v.begin());
  return bit_reverse_rows(dft_batch(padded));
}

// ------------------------------------------------------------------ Merkle MMCS
// [UPSTREAM-RECALL p3-merkle-tree 0.5.1 MerkleTree::new / compress_and_inject / open_batch / verify_batch]
std::vector<Digest> MerkleTree::cap() const {
  size_t L = layers.size();
  size_t ch = std::min<size_t>(cap_height, L - 1);
  return layers[L - 1 - ch];
}

static Digest hash_rows(const std::vector<const Mat*>& group, size_t row) {
  size_t tot = 0;
  for (auto m : group) tot += m->w;
  std::vector<u64> buf;
  buf.reserve(tot);
  for (auto m : group) buf.insert(buf.end(), &m->v[row * m->w], &m->v[row * m->w] + m->w);
  return hash_elems(buf.data(), buf.size());
}

void mmcs_commit(std::vector<Mat>&& mats, unsigned cap_height, MerkleTree& out) {
  out.mats = std::move(mats);
  out.cap_height = cap_height;
  out.layers.clear();
  if (out.mats.empty()) throw std::runtime_error("mmcs_commit: no matrices");
  // stable sort by height, tallest first
  std::vector<const Mat*> order;
  for (auto& m : out.mats) {
    if (m.h == 0 || (m.h & (m.h - 1))) throw std::runtime_error("mmcs_commit: heights must be powers of two");
    order.push_back(&m);
  }
  std::stable_sort(order.begin(), order.end(), [](const Mat* a, const Mat* b) { return a->h > b->h; });
  size_t pos = 0;
  size_t maxh = order[0]->h;
  std::vector<const Mat*> group;
  while (pos < order.size() && order[pos]->h == maxh) group.push_back(order[pos++]);
  std::vector<Digest> layer(maxh);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < maxh; i++) layer[i] = hash_rows(group, i);
  out.layers.push_back(std::move(layer));
  while (out.layers.back().size() > 1) {
    const std::vector<Digest>& prev = out.layers.back();
    size_t nl = prev.size() / 2;
    group.clear();
    while (pos < order.size() && order[pos]->h == nl) group.push_back(order[pos++]);
    std::vector<Digest> next(nl);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < nl; i++) {
      Digest d = compress2(prev[2 * i], prev[2 * i + 1]);
      if (!group.empty()) d = compress2(d, hash_rows(group, i));
      next[i] = d;
    }
    out.layers.push_back(std::move(next));
  }
  if (pos != order.size()) throw std::runtime_error("mmcs_commit: matrix height not reached");
}

BatchOpening mmcs_open_batch(const MerkleTree& t, size_t index) {
  BatchOpening o;
  size_t maxh = t.max_height();
  unsigned log_max = log2_strict(maxh);
  for (auto& m : t.mats) {
    unsigned lh = log2_strict(m.h);
    size_t r = index >> (log_max - lh);
    o.opened_values.emplace_back(&m.v[r * m.w], &m.v[r * m.w] + m.w);
  }
  size_t ch = std::min<size_t>(t.cap_height, t.layers.size() - 1);
  for (size_t i = 0; i + ch < log_max; i++) o.proof.push_back(t.layers[i][(index >> i) ^ 1]);
  return o;
}

bool mmcs_verify_batch(const std::vector<Digest>& cap, const std::vector<Dim>& dims, size_t index,
                       const BatchOpening& opening) {
  if (dims.size() != opening.opened_values.size() || dims.empty()) return false;
  std::vector<size_t> order(dims.size());
  for (size_t i = 0; i < dims.size(); i++) {
    order[i] = i;
    if (opening.opened_values[i].size() != dims[i].w) return false;
    if (dims[i].h == 0 || (dims[i].h & (dims[i].h - 1))) return false;
  }
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return dims[a].h > dims[b].h; });
  size_t pos = 0, cur = dims[order[0]].h;
  unsigned log_max = log2_strict(cur);
  auto hash_group = [&](size_t height) {
    std::vector<u64> buf;
    while (pos < order.size() && dims[order[pos]].h == height) {
      auto& v = opening.opened_values[order[pos]];
      buf.insert(buf.end(), v.begin(), v.end());
      pos++;
    }
    return hash_elems(buf.data(), buf.size());
  };
  Digest root = hash_group(cur);
  // cap must be a power of two; its log is the effective cap height
  size_t capn = cap.size();
  if (capn == 0 || (capn & (capn - 1))) return false;
  unsigned ch = log2_strict(capn);
  if (ch > log_max) return false;
  if (opening.proof.size() != log_max - ch) return false;
  size_t idx = index;
  if (idx >= (size_t(1) << log_max)) return false;
  for (auto& sib : opening.proof) {
    root = (idx & 1) ? compress2(sib, root) : compress2(root, sib);
    idx >>= 1;
    cur >>= 1;
    if (pos < order.size() && dims[order[pos]].h == cur) root = compress2(root, hash_group(cur));
  }
  if (pos != order.size()) return false;
  return root == cap[idx];
}

// ------------------------------------------------------------------ challenger
#ifndef MSO_BABYBEAR
// [UPSTREAM-RECALL p3-challenger 0.5.1 HashChallenger / SerializingChallenger64]
void Challenger::observe(u64 canonical) {
  uint8_t b[8];
  for (int k = 0; k < 8; k++) b[k] = (uint8_t)(canonical >> (8 * k));
  observe_bytes(b, 8);
}
uint8_t Challenger::sample_byte() {
  if (output.empty()) {
    Digest d = hash_bytes(input.data(), input.size());
    input.assign(d.b, d.b + 32);
    output.assign(d.b, d.b + 32);
  }
  uint8_t b = output.back();  // pop from the back
  output.pop_back();
  return b;
}
u64 Challenger::sample_u64() {
  u64 v = 0;
  for (int k = 0; k < 8; k++) v |= (u64)sample_byte() << (8 * k);
  return v;
}
u64 Challenger::sample_base() {
  for (;;) {
    u64 v = sample_u64();
    if (v < F_P) return v;
  }
}
size_t Challenger::sample_bits(unsigned bits) {
  u64 v = sample_u64();
  return (size_t)(v & ((u64(1) << bits) - 1));
}
#else
// [UPSTREAM-RECALL p3-challenger 0.5.1 DuplexChallenger]: observe clears the output buffer and queues the value; a full
// queue (RATE = 8) is absorbed at once: the queued values overwrite the first state words, the state is permuted and
// the first 8 words become the output buffer. sample absorbs whatever is queued (or refills an empty output buffer)
// and pops from the BACK of the output buffer.
void Challenger::duplexing() {
  for (size_t i = 0; i < input.size(); i++) state[i] = input[i];
  input.clear();
  poseidon2_permute(state);
  output.assign(state, state + 8);
}
void Challenger::observe(u64 canonical) {
  output.clear();
  input.push_back(canonical);
  if (input.size() == 8) duplexing();
}
u64 Challenger::sample_base() {
  if (!input.empty() || output.empty()) duplexing();
  u64 v = output.back();
  output.pop_back();
  return v;
}
// sample_bits: the low bits of the canonical value of one sampled element
size_t Challenger::sample_bits(unsigned bits) {
  if (bits >= 31) throw std::runtime_error("sample_bits: too many bits for a 31-bit field");
  return (size_t)(sample_base() & ((u64(1) << bits) - 1));
}
#endif
EF Challenger::sample_ext() {
  EF e;
  for (unsigned k = 0; k < EXT_D; k++) e.c[k] = sample_base();
  return e;
}
bool Challenger::check_witness(unsigned bits, u64 witness) {
  if (bits == 0) return true;
  observe(witness);
  return sample_bits(bits) == 0;
}
u64 Challenger::grind(unsigned bits) {
  if (bits == 0) return 0;  // DeterministicPow, src/types.rs:75-80
  for (u64 w = 0;; w++) {
    if (w >= F_P) throw std::runtime_error("grind: no witness");
    Challenger c = *this;
    if (c.check_witness(bits, w)) {
      check_witness(bits, w);
      return w;
    }
  }
}

// ------------------------------------------------------------------ system
size_t Circuit::quotient_degree() const {
  size_t d = std::max<size_t>(max_constraint_degree, 2) - 1;
  size_t q = 1;
  while (q < d) q <<= 1;
  return q;
}

#ifndef MSO_BABYBEAR
std::vector<uint8_t> System::challenger_seed() const {
  // src/types.rs:118-130
  const char* tag = "multi-stark/v0";
  std::vector<uint8_t> s(tag, tag + 14);
  const u64 ps[7] = {params.log_blowup,  params.cap_height,      params.log_final_poly_len, params.max_log_arity,
                     params.num_queries, params.commit_pow_bits, params.query_pow_bits};
  for (u64 p : ps)
    for (int k = 0; k < 8; k++) s.push_back((uint8_t)(p >> (8 * k)));
  return s;
}
#else
std::vector<u64> System::challenger_seed() const {
  // src/test_circuits/baby_bear_config.rs:72-86: the tag bytes and the seven parameters, each as a field element
  const char* tag = "multi-stark/v0";
  std::vector<u64> s(tag, tag + 14);
  const u64 ps[7] = {params.log_blowup,  params.cap_height,      params.log_final_poly_len, params.max_log_arity,
                     params.num_queries, params.commit_pow_bits, params.query_pow_bits};
  for (u64 p : ps) s.push_back(p % F_P);
  return s;
}
#endif

void System::observe_shape(Challenger& ch) const {
  // src/system.rs:211-222
  ch.observe(f_from_u64((u64)circuits.size()));
  for (auto& c : circuits) {
    ch.observe(f_from_u64((u64)c.constraint_count));
    ch.observe(f_from_u64((u64)c.max_constraint_degree));
    ch.observe(f_from_u64((u64)c.pre_height));
    ch.observe(f_from_u64((u64)c.pre_width));
    ch.observe(f_from_u64((u64)c.main_width));
    ch.observe(f_from_u64((u64)c.stage2_width));
  }
}

namespace {
struct Reader {
  const uint8_t* p;
  size_t n, off = 0;
  u64 word() {
    if (off + 8 > n) throw std::runtime_error("blob truncated");
    u64 v = 0;
    for (int k = 0; k < 8; k++) v |= (u64)p[off + k] << (8 * k);
    off += 8;
    return v;
  }
};
}  // namespace

#ifndef MSO_BABYBEAR
static const u64 BLOB_MAGIC = 0x31305359534D0000ULL;  // "\0\0MSYS01"
#else
static const u64 BLOB_MAGIC = 0x31304259534D0000ULL;  // "\0\0MSYB01": the parameters are followed by the Poseidon2 constants
#endif

System system_from_blob(const uint8_t* blob, size_t len) {
  Reader rd{blob, len};
  if (rd.word() != BLOB_MAGIC) throw std::runtime_error("bad system blob magic");
  System sys;
  Params& p = sys.params;
  p.log_blowup = rd.word();
  p.cap_height = rd.word();
  p.log_final_poly_len = rd.word();
  p.max_log_arity = rd.word();
  p.num_queries = rd.word();
  p.commit_pow_bits = rd.word();
  p.query_pow_bits = rd.word();
  if (p.max_log_arity < 1 || p.max_log_arity > 16) throw std::runtime_error("max_log_arity out of range (1..16)");
  if (p.log_blowup < 1 || p.log_blowup > 8) throw std::runtime_error("bad log_blowup");
#ifdef MSO_BABYBEAR
  {
    Poseidon2Constants& k = poseidon2_constants();
    for (int r = 0; r < 8; r++)
      for (int i = 0; i < 16; i++)
        if ((k.external[r][i] = rd.word()) >= F_P) throw std::runtime_error("non-canonical round constant");
    for (int r = 0; r < 13; r++)
      if ((k.internal[r] = rd.word()) >= F_P) throw std::runtime_error("non-canonical round constant");
  }
#endif
  size_t nc = rd.word();
  const size_t D = EXT_D;
  std::vector<Mat> pre_traces;
  for (size_t ci = 0; ci < nc; ci++) {
    Circuit c;
    c.main_width = rd.word();
    c.pre_width = rd.word();
    c.pre_height = rd.word();
    size_t nn = rd.word(), nz = rd.word(), nl = rd.word();
    c.num_lookups = nl;
    c.stage2_width = std::max<size_t>(nl, 1) * D;  // src/lookup.rs:90-92
    c.num_publics = 4 * D;                          // src/lookup.rs:82-84
    c.nodes.resize(nn);
    c.degrees.resize(nn);
    for (size_t i = 0; i < nn; i++) {
      u64 w0 = rd.word();
      Node& nd = c.nodes[i];
      nd.kind = (uint32_t)(w0 & 0xff);
      nd.source = (uint32_t)((w0 >> 8) & 0xff);
      nd.offset = (uint32_t)((w0 >> 16) & 0xff);
      nd.a = rd.word();
      nd.b = rd.word();
      auto child = [&](u64 id) {
        if (id >= i) throw std::runtime_error("node program not topologically ordered");
        return c.degrees[id];
      };
      uint32_t deg = 0;
      switch (nd.kind) {  // src/graph.rs:242-252
        case N_CONST:
          if (nd.a >= F_P) throw std::runtime_error("non-canonical constant");
          deg = 0;
          break;
        case N_PUBLIC:
          if (nd.a >= c.num_publics) throw std::runtime_error("public out of range");
          deg = 0;
          break;
        case N_IS_TRANS: deg = 0; break;
        case N_VAR: {
          size_t width = nd.source == SRC_PRE ? c.pre_width : nd.source == SRC_MAIN ? c.main_width : c.stage2_width;
          if (nd.source > SRC_STAGE2 || nd.offset > 1 || nd.a >= width) throw std::runtime_error("column out of range");
          deg = 1;
          break;
        }
        case N_IS_FIRST:
        case N_IS_LAST: deg = 1; break;
        case N_ADD:
        case N_SUB: deg = std::max(child(nd.a), child(nd.b)); break;
        case N_MUL: deg = child(nd.a) + child(nd.b); break;
        case N_NEG: deg = child(nd.a); break;
        default: throw std::runtime_error("bad node kind");
      }
      c.degrees[i] = deg;
    }
    c.zeros.resize(nz);
    uint32_t graph_deg = 0;
    for (size_t i = 0; i < nz; i++) {
      u64 z = rd.word();
      if (z >= nn) throw std::runtime_error("zero root out of range");
      c.zeros[i] = (uint32_t)z;
      graph_deg = std::max(graph_deg, c.degrees[z]);
    }
    c.lookups.resize(nl);
    uint32_t logup_deg = nl ? 0 : 1;  // src/lookup.rs:262-278
    for (size_t j = 0; j < nl; j++) {
      Lookup& l = c.lookups[j];
      u64 m = rd.word();
      if (m >= nn) throw std::runtime_error("lookup node out of range");
      l.mult = (uint32_t)m;
      size_t na = rd.word();
      uint32_t msg = 0;
      for (size_t k = 0; k < na; k++) {
        u64 a = rd.word();
        if (a >= nn) throw std::runtime_error("lookup node out of range");
        l.args.push_back((uint32_t)a);
        msg = std::max(msg, c.degrees[a]);
      }
      logup_deg = std::max(logup_deg, std::max(msg + 1, c.degrees[m]));
    }
    c.constraint_count = nz + std::max<size_t>(nl, 1) * D;  // src/system.rs:151
    c.max_constraint_degree = std::max(graph_deg, logup_deg);
    if (c.quotient_degree() > (size_t(1) << p.log_blowup))
      throw std::runtime_error("constraint degree needs a quotient degree beyond the blowup");  // src/system.rs:171-178
    if (c.pre_width) {
      if (c.pre_height == 0 || (c.pre_height & (c.pre_height - 1))) throw std::runtime_error("bad preprocessed height");
      c.preprocessed = Mat(c.pre_height, c.pre_width);
      for (auto& x : c.preprocessed.v) {
        x = rd.word();
        if (x >= F_P) throw std::runtime_error("non-canonical preprocessed value");
      }
      sys.pre_indices.push_back((int)pre_traces.size());
      pre_traces.push_back(coset_lde_bitrev(c.preprocessed, (unsigned)p.log_blowup, F_GENERATOR));
    } else {
      c.pre_height = 0;
      sys.pre_indices.push_back(-1);
    }
    sys.circuits.push_back(std::move(c));
  }
  if (rd.off != len) throw std::runtime_error("trailing bytes in system blob");
  if (!pre_traces.empty()) {
    sys.has_pre = true;
    mmcs_commit(std::move(pre_traces), (unsigned)p.cap_height, sys.pre_tree);
    sys.pre_commit = sys.pre_tree.cap();
  }
  return sys;
}

// ------------------------------------------------------------------ proof bytes
// Proof::to_bytes, src/prover.rs:241-248: bincode 2 standard().with_little_endian().with_fixed_int_encoding()
// over the serde-derived containers. [UPSTREAM-RECALL for the inner p3 containers; field order of
// Proof/Commitments is in-tree src/prover.rs:201-238]. Vec -> u64 length + items; u8/bool -> 1 byte;
// Goldilocks -> u64; Ext2 -> 2 x u64; digest -> 32 raw bytes; Option -> 1-byte tag.
namespace {
struct W {
  std::vector<uint8_t> b;
  void u8(uint8_t x) { b.push_back(x); }
  void u64_(u64 x) {
    for (int k = 0; k < 8; k++) b.push_back((uint8_t)(x >> (8 * k)));
  }
#ifndef MSO_BABYBEAR
  void fe(u64 x) { u64_(x); }
  void dig(const Digest& d) { b.insert(b.end(), d.b, d.b + 32); }
#else
  void fe(u64 x) {  // MontyField31 serialises its Montgomery form as a u32
    uint32_t m = bb_to_wire(x);
    for (int k = 0; k < 4; k++) b.push_back((uint8_t)(m >> (8 * k)));
  }
  void dig(const Digest& d) {  // [BabyBear; 8]: a fixed-size array, no length prefix
    for (int i = 0; i < 8; i++) fe(digest_elem(d, i));
  }
#endif
  void ext(EF e) {
    for (unsigned k = 0; k < EXT_D; k++) fe(e.c[k]);
  }
  void cap(const std::vector<Digest>& c) {
    u64_(c.size());
    for (auto& d : c) dig(d);
  }
  void round(const OpenedRound& r) {
    u64_(r.size());
    for (auto& m : r) {
      u64_(m.size());
      for (auto& pt : m) {
        u64_(pt.size());
        for (auto& e : pt) ext(e);
      }
    }
  }
};
struct R {
  const uint8_t* p;
  size_t n, off = 0;
  void need(size_t k) {
    if (off + k > n) throw std::runtime_error("proof truncated");
  }
  uint8_t u8() {
    need(1);
    return p[off++];
  }
  u64 u64_() {
    need(8);
    u64 v = 0;
    for (int k = 0; k < 8; k++) v |= (u64)p[off + k] << (8 * k);
    off += 8;
    return v;
  }
  size_t len(size_t item_min) {
    u64 l = u64_();
    if (item_min && l > (n - off) / item_min) throw std::runtime_error("proof length field too large");
    return (size_t)l;
  }
#ifndef MSO_BABYBEAR
  u64 fe() {
    u64 v = u64_();
    if (v >= F_P) throw std::runtime_error("non-canonical field element");
    return v;
  }
  Digest dig() {
    need(32);
    Digest d;
    memcpy(d.b, p + off, 32);
    off += 32;
    return d;
  }
#else
  u64 fe() {
    need(4);
    uint32_t m = 0;
    for (int k = 0; k < 4; k++) m |= (uint32_t)p[off + k] << (8 * k);
    off += 4;
    if (m >= F_P) throw std::runtime_error("non-canonical field element");
    return bb_from_wire(m);
  }
  Digest dig() {
    Digest d;
    for (int i = 0; i < 8; i++) digest_set(d, i, fe());
    return d;
  }
#endif
  EF ext() {
    EF e;
    for (unsigned k = 0; k < EXT_D; k++) e.c[k] = fe();
    return e;
  }
  std::vector<Digest> cap() {
    size_t l = len(32);
    std::vector<Digest> c(l);
    for (auto& d : c) d = dig();
    return c;
  }
  OpenedRound round() {
    OpenedRound r(len(8));
    for (auto& m : r) {
      m.resize(len(8));
      for (auto& pt : m) {
        pt.resize(len(EXT_D * F_WIRE_BYTES));
        for (auto& e : pt) e = ext();
      }
    }
    return r;
  }
};
}  // namespace

static void write_fri(W& w, const FriProof& f) {
  w.u64_(f.commit_phase_commits.size());
  for (auto& c : f.commit_phase_commits) w.cap(c);
  w.u64_(f.commit_pow_witnesses.size());
  for (auto x : f.commit_pow_witnesses) w.fe(x);
  w.u64_(f.query_proofs.size());
  for (auto& q : f.query_proofs) {
    w.u64_(q.input_proof.size());
    for (auto& bo : q.input_proof) {
      w.u64_(bo.opened_values.size());
      for (auto& row : bo.opened_values) {
        w.u64_(row.size());
        for (auto x : row) w.fe(x);
      }
      w.u64_(bo.proof.size());
      for (auto& d : bo.proof) w.dig(d);
    }
    w.u64_(q.commit_phase_openings.size());
    for (auto& s : q.commit_phase_openings) {
      w.u8(s.log_arity);
      w.u64_(s.sibling_values.size());
      for (auto& e : s.sibling_values) w.ext(e);
      w.u64_(s.proof.size());
      for (auto& d : s.proof) w.dig(d);
    }
  }
  w.u64_(f.final_poly.size());
  for (auto& e : f.final_poly) w.ext(e);
  w.fe(f.query_pow_witness);
}

std::vector<uint8_t> fri_to_bytes(const FriProof& f) {
  W w;
  write_fri(w, f);
  return std::move(w.b);
}

std::vector<uint8_t> proof_to_bytes(const Proof& p) {
  W w;
  w.u64_(p.active.size());
  for (auto a : p.active) w.u8(a ? 1 : 0);
  w.cap(p.stage1_commit);
  w.cap(p.stage2_commit);
  w.cap(p.quotient_commit);
  w.u64_(p.intermediate_accumulators.size());
  for (auto& e : p.intermediate_accumulators) w.ext(e);
  w.u64_(p.log_degrees.size());
  for (auto d : p.log_degrees) w.u8(d);
  write_fri(w, p.opening_proof);
  w.round(p.quotient_opened);
  w.u8(p.has_pre_opened ? 1 : 0);
  if (p.has_pre_opened) w.round(p.pre_opened);
  w.round(p.stage1_opened);
  w.round(p.stage2_opened);
  return std::move(w.b);
}

static FriProof read_fri(R& r) {
  FriProof f;
  f.commit_phase_commits.resize(r.len(8));
  for (auto& c : f.commit_phase_commits) c = r.cap();
  f.commit_pow_witnesses.resize(r.len(F_WIRE_BYTES));
  for (auto& x : f.commit_pow_witnesses) x = r.fe();
  f.query_proofs.resize(r.len(16));
  for (auto& q : f.query_proofs) {
    q.input_proof.resize(r.len(16));
    for (auto& bo : q.input_proof) {
      bo.opened_values.resize(r.len(8));
      for (auto& row : bo.opened_values) {
        row.resize(r.len(F_WIRE_BYTES));
        for (auto& x : row) x = r.fe();
      }
      bo.proof.resize(r.len(32));
      for (auto& d : bo.proof) d = r.dig();
    }
    q.commit_phase_openings.resize(r.len(17));
    for (auto& s : q.commit_phase_openings) {
      s.log_arity = r.u8();
      s.sibling_values.resize(r.len(EXT_D * F_WIRE_BYTES));
      for (auto& e : s.sibling_values) e = r.ext();
      s.proof.resize(r.len(32));
      for (auto& d : s.proof) d = r.dig();
    }
  }
  f.final_poly.resize(r.len(EXT_D * F_WIRE_BYTES));
  for (auto& e : f.final_poly) e = r.ext();
  f.query_pow_witness = r.fe();
  return f;
}

FriProof fri_from_bytes(const uint8_t* bytes, size_t n) {
  R r{bytes, n};
  FriProof f = read_fri(r);
  if (r.off != n) throw std::runtime_error("trailing bytes in FRI proof");
  return f;
}

Proof proof_from_bytes(const uint8_t* bytes, size_t n) {
  R r{bytes, n};
  Proof p;
  p.active.resize(r.len(1));
  for (auto& a : p.active) {
    a = r.u8();
    if (a > 1) throw std::runtime_error("bad bool");
  }
  p.stage1_commit = r.cap();
  p.stage2_commit = r.cap();
  p.quotient_commit = r.cap();
  p.intermediate_accumulators.resize(r.len(EXT_D * F_WIRE_BYTES));
  for (auto& e : p.intermediate_accumulators) e = r.ext();
  p.log_degrees.resize(r.len(1));
  for (auto& d : p.log_degrees) d = r.u8();
  p.opening_proof = read_fri(r);
  p.quotient_opened = r.round();
  uint8_t tag = r.u8();
  if (tag > 1) throw std::runtime_error("bad option tag");
  p.has_pre_opened = tag == 1;
  if (p.has_pre_opened) p.pre_opened = r.round();
  p.stage1_opened = r.round();
  p.stage2_opened = r.round();
  if (r.off != n) throw std::runtime_error("trailing bytes in proof");
  return p;
}

}  // namespace mso
