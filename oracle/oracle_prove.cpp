// ORACLE — test infrastructure only (see ms_oracle.hpp header). Witness preparation, stage-2 traces,
// quotient evaluation, PCS open / FRI prover, prove() and verify().
#include <algorithm>
#include <chrono>
#include <cstring>
#include <map>

#include "ms_oracle.hpp"

namespace mso {

// ------------------------------------------------------------------ generic sweep (src/eval.rs:36-111)
namespace {
struct BaseOps {
  typedef u64 T;
  static T from_base(u64 c) { return c; }
  static T add(T a, T b) { return f_add(a, b); }
  static T sub(T a, T b) { return f_sub(a, b); }
  static T mul(T a, T b) { return f_mul(a, b); }
  static T neg(T a) { return f_neg(a); }
  static T mul_base(T a, u64 b) { return f_mul(a, b); }
  static T zero() { return 0; }
};
struct ExtOps {
  typedef EF T;
  static T from_base(u64 c) { return ef(c); }
  static T add(T a, T b) { return ef_add(a, b); }
  static T sub(T a, T b) { return ef_sub(a, b); }
  static T mul(T a, T b) { return ef_mul(a, b); }
  static T neg(T a) { return ef_neg(a); }
  static T mul_base(T a, u64 b) { return ef_mul_base(a, b); }
  static T zero() { return ef(0); }
};

template <class O>
struct View {
  typedef typename O::T T;
  const T* pre[2];
  const T* main[2];
  const T* s2[2];
  const T* publics;
  T is_first, is_last, is_trans;
};

template <class O>
void sweep(const Circuit& c, const View<O>& v, std::vector<typename O::T>& buf, size_t len) {
  typedef typename O::T T;
  buf.resize(len);
  for (size_t i = 0; i < len; i++) {
    const Node& n = c.nodes[i];
    T val;
    switch (n.kind) {
      case N_CONST: val = O::from_base(n.a); break;
      case N_VAR: {
        const T* const* rows = n.source == SRC_PRE ? v.pre : n.source == SRC_MAIN ? v.main : v.s2;
        val = rows[n.offset][n.a];
        break;
      }
      case N_PUBLIC: val = v.publics[n.a]; break;
      case N_IS_FIRST: val = v.is_first; break;
      case N_IS_LAST: val = v.is_last; break;
      case N_IS_TRANS: val = v.is_trans; break;
      case N_ADD: val = O::add(buf[n.a], buf[n.b]); break;
      case N_SUB: val = O::sub(buf[n.a], buf[n.b]); break;
      case N_MUL: val = O::mul(buf[n.a], buf[n.b]); break;
      default: val = O::neg(buf[n.a]); break;
    }
    buf[i] = val;
  }
}

// src/lookup.rs:103-128: coordinate product in X^D = W on D coordinates of the working type (the values are the
// coordinates of the extension product, whichever multiplication schedule - schoolbook or the D = 2 Karatsuba - is used)
template <class O>
inline void coord_mul(const typename O::T* a, const typename O::T* b, typename O::T* out) {
  typedef typename O::T T;
  T lo[EXT_D], hi[EXT_D];
  for (unsigned k = 0; k < EXT_D; k++) lo[k] = hi[k] = O::zero();
  for (unsigned i = 0; i < EXT_D; i++)
    for (unsigned j = 0; j < EXT_D; j++) {
      T p = O::mul(a[i], b[j]);
      if (i + j < EXT_D)
        lo[i + j] = O::add(lo[i + j], p);
      else
        hi[i + j - EXT_D] = O::add(hi[i + j - EXT_D], p);
    }
  for (unsigned k = 0; k < EXT_D; k++) out[k] = O::add(lo[k], O::mul_base(hi[k], EXT_W));
}

// src/lookup.rs:152-256
template <class O>
void logup_constraint_values(const Circuit& c, const std::vector<typename O::T>& nv, const typename O::T* s2,
                             const typename O::T* s2n, const typename O::T* publics,
                             const typename O::T* delta_scaled, typename O::T is_last,
                             std::vector<typename O::T>& out) {
  typedef typename O::T T;
  const unsigned D = EXT_D;
  const T* beta = publics;
  const T* gamma = publics + D;
  T inj[EXT_D];
  for (unsigned k = 0; k < D; k++) inj[k] = O::mul(is_last, delta_scaled[k]);
  if (c.lookups.empty()) {
    for (unsigned k = 0; k < D; k++) out.push_back(O::add(O::sub(s2n[k], s2[k]), inj[k]));
    return;
  }
  size_t last = c.lookups.size() - 1;
  for (size_t j = 0; j < c.lookups.size(); j++) {
    const Lookup& l = c.lookups[j];
    T diff[EXT_D];
    for (unsigned k = 0; k < D; k++) {
      T tgt = j < last ? s2[D * (j + 1) + k] : O::add(s2n[k], inj[k]);
      diff[k] = O::sub(tgt, s2[D * j + k]);
    }
    T f[EXT_D], g[EXT_D];
    for (unsigned k = 0; k < D; k++) f[k] = O::zero();
    for (size_t a = l.args.size(); a-- > 0;) {
      coord_mul<O>(f, gamma, g);
      for (unsigned k = 0; k < D; k++) f[k] = g[k];
      f[0] = O::add(f[0], nv[l.args[a]]);
    }
    for (unsigned k = 0; k < D; k++) f[k] = O::add(f[k], beta[k]);
    coord_mul<O>(f, diff, g);
    out.push_back(O::sub(g[0], nv[l.mult]));
    for (unsigned k = 1; k < D; k++) out.push_back(g[k]);
  }
}

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

// ------------------------------------------------------------------ witness (src/system.rs:244-328)
static size_t lookup_prefix_len(const Circuit& c) {
  // everything reachable from the lookup expressions lives in a prefix (src/graph.rs:126-136); the prefix
  // end is one past the largest lookup node id because children precede parents.
  size_t len = 0;
  for (auto& l : c.lookups) {
    len = std::max<size_t>(len, l.mult + 1);
    for (auto a : l.args) len = std::max<size_t>(len, a + 1);
  }
  return len;
}

Witness witness_from_stage_1(const System& sys, std::vector<Mat>&& traces) {
  if (traces.size() != sys.circuits.size()) throw std::runtime_error("expected one trace per circuit");
  Witness w;
  w.lookups.resize(traces.size());
  for (size_t ci = 0; ci < traces.size(); ci++) {
    const Circuit& c = sys.circuits[ci];
    const Mat& tr = traces[ci];
    if (tr.h && tr.w != c.main_width) throw std::runtime_error("trace width mismatch");
    if (c.pre_width && tr.h && tr.h != c.pre_height)
      throw std::runtime_error("main trace height must equal preprocessed trace height");
    LookupValues& lv = w.lookups[ci];
    lv.height = tr.h;
    lv.num_lookups = c.lookups.size();
    lv.arg_offsets.push_back(0);
    for (auto& l : c.lookups) lv.arg_offsets.push_back(lv.arg_offsets.back() + l.args.size());
    size_t aw = lv.arg_offsets.back();
    if (tr.h == 0 || c.lookups.empty()) continue;
    lv.mult.assign(tr.h * lv.num_lookups, 0);
    lv.args.assign(tr.h * aw, 0);
    size_t plen = lookup_prefix_len(c);
#pragma omp parallel
    {
      std::vector<u64> buf;
#pragma omp for schedule(static)
      for (size_t r = 0; r < tr.h; r++) {
        size_t rn = (r + 1) % tr.h;
        View<BaseOps> v;
        v.pre[0] = c.pre_width ? &c.preprocessed.v[r * c.pre_width] : nullptr;
        v.pre[1] = c.pre_width ? &c.preprocessed.v[rn * c.pre_width] : nullptr;
        v.main[0] = &tr.v[r * tr.w];
        v.main[1] = &tr.v[rn * tr.w];
        v.s2[0] = v.s2[1] = nullptr;
        v.publics = nullptr;
        v.is_first = r == 0;
        v.is_last = r == tr.h - 1;
        v.is_trans = r != tr.h - 1;
        sweep<BaseOps>(c, v, buf, plen);
        for (size_t j = 0; j < c.lookups.size(); j++) {
          lv.mult[r * lv.num_lookups + j] = buf[c.lookups[j].mult];
          for (size_t k = 0; k < c.lookups[j].args.size(); k++)
            lv.args[r * aw + lv.arg_offsets[j] + k] = buf[c.lookups[j].args[k]];
        }
      }
    }
  }
  w.traces = std::move(traces);
  return w;
}

// ------------------------------------------------------------------ lookups
// src/lookup.rs:375-384: Horner over the reversed coefficients
EF fingerprint(EF r, const u64* coeffs, size_t n) {
  EF acc = ef(0);
  for (size_t k = n; k-- > 0;) acc = ef_add(ef_mul(acc, r), ef(coeffs[k]));
  return acc;
}

// src/prover.rs:382-387
EF claims_accumulator(const std::vector<std::vector<u64>>& claims, EF beta, EF gamma) {
  EF acc = ef(0);
  // exact field sum: order of additions is irrelevant, so the loop may be chunked
  size_t n = claims.size();
  std::vector<EF> part;
#pragma omp parallel
  {
    EF local = ef(0);
#pragma omp for schedule(static) nowait
    for (size_t i = 0; i < n; i++) {
      EF m = ef_add(beta, fingerprint(gamma, claims[i].data(), claims[i].size()));
      local = ef_add(local, ef_inv(m));
    }
#pragma omp critical
    acc = ef_add(acc, local);
  }
  return acc;
}

// p3_field::batch_multiplicative_inverse (values are mathematically determined)
static void batch_inverse(std::vector<EF>& v) {
  size_t n = v.size();
  const size_t CH = 1024;
#pragma omp parallel for schedule(static)
  for (size_t s = 0; s < n; s += CH) {
    size_t e = std::min(n, s + CH);
    std::vector<EF> pre(e - s);
    EF acc = ef(1);
    for (size_t i = s; i < e; i++) {
      pre[i - s] = acc;
      acc = ef_mul(acc, v[i]);
    }
    EF inv = ef_inv(acc);
    for (size_t i = e; i-- > s;) {
      EF x = v[i];
      v[i] = ef_mul(inv, pre[i - s]);
      inv = ef_mul(inv, x);
    }
  }
}

// src/lookup.rs:472-555
void stage_2_traces(const std::vector<LookupValues>& circuits, EF beta, EF gamma, EF accumulator,
                    std::vector<std::vector<EF>>& traces, std::vector<EF>& accs) {
  traces.clear();
  accs.clear();
  for (auto& c : circuits) {
    size_t nm = c.height * c.num_lookups;
    std::vector<EF> msgs(nm);
    size_t aw = c.arg_offsets.empty() ? 0 : c.arg_offsets.back();
#pragma omp parallel for schedule(static)
    for (size_t idx = 0; idx < nm; idx++) {
      size_t row = idx / c.num_lookups, l = idx % c.num_lookups;
      const u64* a = c.args.data() + row * aw + c.arg_offsets[l];
      msgs[idx] = ef_add(beta, fingerprint(gamma, a, c.arg_offsets[l + 1] - c.arg_offsets[l]));
    }
    batch_inverse(msgs);
    std::vector<EF> vec;
    if (c.num_lookups == 0) {
      vec.assign(c.height, ef(0));
    } else {
      vec.resize(nm);
      EF local = ef(0);
      for (size_t idx = 0; idx < nm; idx++) {  // serial, as in the reference (src/lookup.rs:530-543)
        vec[idx] = local;
        local = ef_add(local, ef_mul_base(msgs[idx], c.mult[idx]));
      }
      accumulator = ef_add(accumulator, local);
    }
    accs.push_back(accumulator);
    traces.push_back(std::move(vec));
  }
}

// ------------------------------------------------------------------ selectors
// [UPSTREAM-RECALL p3-field TwoAdicMultiplicativeCoset::selectors_on_coset; normalisation pinned in-tree by
// src/lookup.rs:697-756]
Selectors selectors_on_coset(unsigned log_n, unsigned log_q) {
  size_t n = size_t(1) << log_n, q = size_t(1) << log_q, N = n * q;
  u64 s_pow_n = f_exp_pow2(F_GENERATOR, log_n);
  std::vector<u64> zh(q), zh_inv(q);
  u64 wq = f_two_adic_generator(log_q), x = 1;
  for (size_t j = 0; j < q; j++) {
    zh[j] = f_sub(f_mul(s_pow_n, x), 1);
    zh_inv[j] = f_inv(zh[j]);
    x = f_mul(x, wq);
  }
  u64 g_inv = f_inv(f_two_adic_generator(log_n));
  u64 wN = f_two_adic_generator(log_n + log_q);
  Selectors s;
  s.is_first.resize(N);
  s.is_last.resize(N);
  s.is_trans.resize(N);
  s.inv_van.resize(N);
  const size_t CH = 4096;
#pragma omp parallel for schedule(static)
  for (size_t st = 0; st < N; st += CH) {
    size_t en = std::min(N, st + CH);
    u64 xi = f_mul(F_GENERATOR, f_pow(wN, st));
    for (size_t i = st; i < en; i++) {
      s.is_first[i] = f_mul(zh[i % q], f_inv(f_sub(xi, 1)));
      s.is_last[i] = f_mul(zh[i % q], f_inv(f_sub(xi, g_inv)));
      s.is_trans[i] = f_sub(xi, g_inv);
      s.inv_van[i] = zh_inv[i % q];
      xi = f_mul(xi, wN);
    }
  }
  return s;
}

// ------------------------------------------------------------------ quotient (src/prover.rs:756-962)
std::vector<EF> quotient_values(const Circuit& c, const u64* publics, unsigned log_n, unsigned log_q,
                                const Mat* pre_q, const Mat& s1_q, const Mat& s2_q, EF alpha) {
  size_t n = size_t(1) << log_n, q = size_t(1) << log_q, N = n * q;
  Selectors sels = selectors_on_coset(log_n, log_q);
  u64 g = f_two_adic_generator(log_n);
  u64 inj_norm = f_inv(f_mul((u64)n % F_P, g));
  size_t next_step = q;
  size_t k = c.constraint_count;
  // reversed alpha powers, src/prover.rs:798-808
  std::vector<EF> apow(k);
  EF a = ef(1);
  for (size_t i = 0; i < k; i++) {
    apow[k - 1 - i] = a;
    a = ef_mul(a, alpha);
  }
  u64 delta_scaled[EXT_D];  // (acc_final - acc_initial) / (n g): publics = beta, gamma, acc_initial, acc_final
  for (unsigned d = 0; d < EXT_D; d++) delta_scaled[d] = f_mul(f_sub(publics[3 * EXT_D + d], publics[2 * EXT_D + d]), inj_norm);
  std::vector<EF> out(N);
#pragma omp parallel
  {
    std::vector<u64> buf, cv;
#pragma omp for schedule(static)
    for (size_t i = 0; i < N; i++) {
      size_t in = (i + next_step) % N;
      View<BaseOps> v;
      v.pre[0] = pre_q ? &pre_q->v[i * pre_q->w] : nullptr;
      v.pre[1] = pre_q ? &pre_q->v[in * pre_q->w] : nullptr;
      v.main[0] = &s1_q.v[i * s1_q.w];
      v.main[1] = &s1_q.v[in * s1_q.w];
      v.s2[0] = &s2_q.v[i * s2_q.w];
      v.s2[1] = &s2_q.v[in * s2_q.w];
      v.publics = publics;
      v.is_first = sels.is_first[i];
      v.is_last = sels.is_last[i];
      v.is_trans = sels.is_trans[i];
      sweep<BaseOps>(c, v, buf, c.nodes.size());
      cv.clear();
      for (auto z : c.zeros) cv.push_back(buf[z]);
      logup_constraint_values<BaseOps>(c, buf, v.s2[0], v.s2[1], publics, delta_scaled, v.is_last, cv);
      EF acc = ef(0);
      for (size_t j = 0; j < k; j++) acc = ef_add(acc, ef_mul_base(apow[j], cv[j]));
      out[i] = ef_mul_base(acc, sels.inv_van[i]);
    }
  }
  return out;
}

// ------------------------------------------------------------------ PCS open + FRI prover
// [UPSTREAM-RECALL p3-fri 0.5.1 TwoAdicFriPcs::open, prover::prove_fri / commit_phase / answer_query,
//  TwoAdicFriFolding::fold_matrix; p3-interpolation interpolate_coset]. Each protocol choice is one function.
namespace {
struct OpenRound {
  const MerkleTree* tree;
  std::vector<std::vector<EF>> points;  // per matrix
};

struct E2Less {
  bool operator()(const EF& a, const EF& b) const { return ef_less(a, b); }
};

// x_i = GENERATOR * w_H^{bitrev(i)}; coset[..2^k] is the bit-reversed coset of size 2^k
std::vector<u64> bitrev_coset(unsigned log_h) {
  size_t H = size_t(1) << log_h;
  std::vector<u64> xs(H);
  u64 w = f_two_adic_generator(log_h);
  const size_t CH = 4096;
#pragma omp parallel for schedule(static)
  for (size_t st = 0; st < H; st += CH) {
    size_t en = std::min(H, st + CH);
    u64 x = f_mul(F_GENERATOR, f_pow(w, st));
    for (size_t i = st; i < en; i++) {
      xs[bitrev(i, log_h)] = x;
      x = f_mul(x, w);
    }
  }
  return xs;
}

// FRI fold of one layer: rows (lo, hi) = evaluations at (x, -x) with x = w_{2R}^{bitrev(i)} (no coset shift:
// p3 folds over the subgroup), result (lo+hi)/2 + beta (lo-hi)/(2x).
std::vector<EF> fri_fold_matrix(EF beta, const std::vector<EF>& cur) {
  size_t rows = cur.size() / 2;
  unsigned lr = log2_strict(rows);
  u64 g_inv = f_inv(f_two_adic_generator(lr + 1));
  u64 half = f_inv(2);
  EF half_beta = ef_mul_base(beta, half);
  std::vector<EF> out(rows);
  const size_t CH = 4096;
#pragma omp parallel for schedule(static)
  for (size_t st = 0; st < rows; st += CH) {
    size_t en = std::min(rows, st + CH);
    u64 gp = f_pow(g_inv, st);
    for (size_t j = st; j < en; j++) {
      // power index j (natural) lands at row bitrev(j)
      size_t i = bitrev(j, lr);
      EF pw = ef_mul_base(half_beta, gp);
      EF lo = cur[2 * i], hi = cur[2 * i + 1];
      out[i] = ef_add(ef_mul(ef_add(ef(half), pw), lo), ef_mul(ef_sub(ef(half), pw), hi));
      gp = f_mul(gp, g_inv);
    }
  }
  return out;
}

// roll-in factor for a reduced opening of matching height: the fold of a round of arity 2^a combines the 2^a
// interleaved parts with beta^0 .. beta^(2^a - 1); the rolled-in vector takes the next power, beta^(2^a) (beta^2 when binary)
EF fri_roll_in_factor(EF beta, unsigned log_arity) { return ef_exp_pow2(beta, log_arity); }

// [UPSTREAM-RECALL p3-fri compute_log_arity_for_round] a round folds as far as max_log_arity allows without stepping over
// the next input's height or below the final height (src/types.rs:189-190: "Maximum folding arity per FRI round (log2)")
unsigned fri_log_arity_for_round(unsigned log_height, int next_input_log_height, unsigned log_final_height, unsigned max_log_arity) {
  unsigned a = std::min(max_log_arity, log_height - log_final_height);
  if (next_input_log_height >= 0) a = std::min(a, log_height - (unsigned)next_input_log_height);
  return a;
}

// A round of arity 2^a: row r of the committed matrix holds storage elements [r 2^a, (r+1) 2^a) - the evaluations over the
// coset x <w_{2^a}> in bit-reversed order - and folds to the value at beta of the polynomial of degree < 2^a through them,
// which is `a` binary folds with beta, beta^2, beta^4, ... (f = sum_j x^j f_j(x^(2^a)) -> sum_j beta^j f_j)
std::vector<EF> fri_fold_matrix_arity(EF beta, unsigned log_arity, std::vector<EF> cur) {
  EF bp = beta;
  for (unsigned j = 0; j < log_arity; j++) {
    cur = fri_fold_matrix(bp, cur);
    bp = ef_square(bp);
  }
  return cur;
}

void pcs_open(const Params& prm, const std::vector<OpenRound>& rounds, Challenger& ch,
              std::vector<OpenedRound>& opened, FriProof& proof) {
  unsigned lb = (unsigned)prm.log_blowup;
  size_t gmax = 0;
  for (auto& r : rounds)
    for (auto& m : r.tree->mats) gmax = std::max(gmax, m.h);
  if (!gmax) throw std::runtime_error("no matrices supplied");
  unsigned log_gmax = log2_strict(gmax);
  std::vector<u64> coset = bitrev_coset(log_gmax);

  // inverse denominators 1/(z - x) per unique point, for the largest height opened there
  std::map<EF, size_t, E2Less> max_h;
  for (auto& r : rounds)
    for (size_t mi = 0; mi < r.tree->mats.size(); mi++)
      for (auto& z : r.points[mi]) {
        size_t& h = max_h[z];
        h = std::max(h, r.tree->mats[mi].h);
      }
  std::map<EF, std::vector<EF>, E2Less> inv_denoms;
  for (auto& kv : max_h) {
    std::vector<EF> d(kv.second);
    EF z = kv.first;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < d.size(); i++) d[i] = ef_sub(z, ef(coset[i]));
    batch_inverse(d);
    inv_denoms.emplace(z, std::move(d));
  }

  // opened values by barycentric interpolation over the first h = height/B storage rows (coset 7*H_h)
  opened.clear();
  for (auto& r : rounds) {
    OpenedRound orr;
    for (size_t mi = 0; mi < r.tree->mats.size(); mi++) {
      const Mat& mat = r.tree->mats[mi];
      size_t h = mat.h >> lb;
      unsigned log_h = log2_strict(h);
      std::vector<std::vector<EF>> per_point;
      for (auto& z : r.points[mi]) {
        const std::vector<EF>& dinv = inv_denoms.at(z);
        std::vector<EF> sums(mat.w, ef(0));
#pragma omp parallel
        {
          std::vector<EF> loc(mat.w, ef(0));
#pragma omp for schedule(static) nowait
          for (size_t i = 0; i < h; i++) {
            EF cs = ef_mul_base(dinv[i], coset[i]);
            const u64* row = &mat.v[i * mat.w];
            for (size_t c = 0; c < mat.w; c++) loc[c] = ef_add(loc[c], ef_mul_base(cs, row[c]));
          }
#pragma omp critical
          for (size_t c = 0; c < mat.w; c++) sums[c] = ef_add(sums[c], loc[c]);
        }
        u64 s_pow = f_exp_pow2(F_GENERATOR, log_h);
        EF vanish = ef_sub(ef_exp_pow2(z, log_h), ef(s_pow));
        u64 denom = f_mul(s_pow, (u64)h % F_P);
        EF scale = ef_mul_base(vanish, f_inv(denom));
        for (auto& y : sums) {
          y = ef_mul(y, scale);
          ch.observe_ext(y);
        }
        per_point.push_back(std::move(sums));
      }
      orr.push_back(std::move(per_point));
    }
    opened.push_back(std::move(orr));
  }

  EF alpha = ch.sample_ext();
  size_t gw = 0;
  for (auto& r : rounds)
    for (auto& m : r.tree->mats) gw = std::max(gw, m.w);
  std::vector<EF> apow(gw + 1);
  apow[0] = ef(1);
  for (size_t i = 1; i <= gw; i++) apow[i] = ef_mul(apow[i - 1], alpha);

  std::vector<size_t> num_reduced(33, 0);
  std::vector<std::vector<EF>> reduced(33);
  std::vector<bool> present(33, false);
  for (size_t ri = 0; ri < rounds.size(); ri++) {
    auto& r = rounds[ri];
    for (size_t mi = 0; mi < r.tree->mats.size(); mi++) {
      const Mat& mat = r.tree->mats[mi];
      unsigned lh = log2_strict(mat.h);
      if (!present[lh]) {
        present[lh] = true;
        reduced[lh].assign(mat.h, ef(0));
      }
      if (r.points[mi].empty()) continue;
      std::vector<EF> comp(mat.h);
#pragma omp parallel for schedule(static)
      for (size_t i = 0; i < mat.h; i++) {
        EF s = ef(0);
        const u64* row = &mat.v[i * mat.w];
        for (size_t c = 0; c < mat.w; c++) s = ef_add(s, ef_mul_base(apow[c], row[c]));
        comp[i] = s;
      }
      for (size_t pi = 0; pi < r.points[mi].size(); pi++) {
        const EF z = r.points[mi][pi];
        const std::vector<EF>& ys = opened[ri][mi][pi];
        EF off = ef_pow(alpha, num_reduced[lh]);
        EF red_z = ef(0);
        for (size_t c = 0; c < mat.w; c++) red_z = ef_add(red_z, ef_mul(apow[c], ys[c]));
        const std::vector<EF>& dinv = inv_denoms.at(z);
        std::vector<EF>& ro = reduced[lh];
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < mat.h; i++)
          ro[i] = ef_add(ro[i], ef_mul(ef_mul(off, ef_sub(red_z, comp[i])), dinv[i]));
        num_reduced[lh] += mat.w;
      }
    }
  }
  std::vector<std::vector<EF>> inputs;
  for (int lh = 32; lh >= 0; lh--)
    if (present[lh]) inputs.push_back(std::move(reduced[lh]));

  // ---- prove_fri: commit phase
  size_t final_len = size_t(1) << prm.log_final_poly_len;
  size_t stop = (size_t(1) << lb) * final_len;
  std::vector<MerkleTree> fri_trees;
  std::vector<EF> folded = std::move(inputs[0]);
  size_t next_in = 1;
  unsigned log_max_height = log2_strict(folded.size());
  proof = FriProof();
  // [UPSTREAM-RECALL p3-fri prove_fri] with log_final_poly_len > 0 the prover asserts that the SHORTEST input is taller
  // than blowup * final length (a shorter one could never be folded into the final polynomial); refuse likewise
  if (prm.log_final_poly_len > 0) {
    size_t min_h = folded.size();
    for (size_t k = 1; k < inputs.size(); k++) min_h = std::min(min_h, inputs[k].size());
    if (min_h <= stop) throw std::runtime_error("FRI: a committed matrix is not taller than blowup * final polynomial length");
  }
  const unsigned log_final_height = lb + (unsigned)prm.log_final_poly_len;
  std::vector<unsigned> arities;
  while (folded.size() > stop) {
    const unsigned la = fri_log_arity_for_round(log2_strict(folded.size()), next_in < inputs.size() ? (int)log2_strict(inputs[next_in].size()) : -1,
                                                log_final_height, (unsigned)prm.max_log_arity);
    arities.push_back(la);
    const size_t ar = size_t(1) << la;
    size_t rows = folded.size() >> la;
    Mat leaves(rows, ar * EXT_D);  // ExtensionMmcs: width-2^a extension rows flattened to 2^a D base columns
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < rows * ar; i++) ef_to(folded[i], &leaves.v[EXT_D * i]);
    fri_trees.emplace_back();
    std::vector<Mat> one;
    one.push_back(std::move(leaves));
    mmcs_commit(std::move(one), (unsigned)prm.cap_height, fri_trees.back());
    std::vector<Digest> cap = fri_trees.back().cap();
    ch.observe_cap(cap);
    proof.commit_phase_commits.push_back(cap);
    proof.commit_pow_witnesses.push_back(ch.grind((unsigned)prm.commit_pow_bits));
    EF beta = ch.sample_ext();
    folded = fri_fold_matrix_arity(beta, la, std::move(folded));
    if (next_in < inputs.size() && inputs[next_in].size() == folded.size()) {
      EF f = fri_roll_in_factor(beta, la);
      const std::vector<EF>& in = inputs[next_in++];
#pragma omp parallel for schedule(static)
      for (size_t i = 0; i < folded.size(); i++) folded[i] = ef_add(folded[i], ef_mul(f, in[i]));
    }
  }
  if (next_in != inputs.size()) throw std::runtime_error("FRI: an input was never rolled in");
  // final polynomial: truncate, undo the bit reversal, inverse DFT
  folded.resize(final_len);
  {
    unsigned lf = (unsigned)prm.log_final_poly_len;
    Mat m(final_len, EXT_D);
    for (size_t i = 0; i < final_len; i++) ef_to(folded[i], &m.v[EXT_D * bitrev(i, lf)]);
    Mat c = idft_batch(m);
    proof.final_poly.resize(final_len);
    for (size_t i = 0; i < final_len; i++) {
      proof.final_poly[i] = ef_from(&c.v[EXT_D * i]);
      ch.observe_ext(proof.final_poly[i]);
    }
  }
  proof.query_pow_witness = ch.grind((unsigned)prm.query_pow_bits);
  // ---- query phase
  for (size_t qi = 0; qi < prm.num_queries; qi++) {
    size_t index = ch.sample_bits(log_max_height);
    QueryProof qp;
    for (auto& r : rounds) {
      unsigned lmh = log2_strict(r.tree->max_height());
      qp.input_proof.push_back(mmcs_open_batch(*r.tree, index >> (log_gmax - lmh)));
    }
    size_t index_i = index;
    for (size_t i = 0; i < fri_trees.size(); i++) {
      const unsigned la = arities[i];
      const size_t ar = size_t(1) << la, own = index_i & (ar - 1), row_i = index_i >> la;
      BatchOpening bo = mmcs_open_batch(fri_trees[i], row_i);
      CommitPhaseStep st;
      st.log_arity = (uint8_t)la;
      const std::vector<u64>& row = bo.opened_values[0];
      for (size_t j = 0; j < ar; j++)  // the row without the queried position's own value, in row order
        if (j != own) st.sibling_values.push_back(ef_from(&row[EXT_D * j]));
      st.proof = std::move(bo.proof);
      qp.commit_phase_openings.push_back(std::move(st));
      index_i = row_i;
    }
    proof.query_proofs.push_back(std::move(qp));
  }
}

Mat evaluations_on_domain(const Mat& lde, size_t size) {
  // p3 get_evaluations_on_domain: first `size` storage rows, viewed bit-reversed (natural order)
  Mat head(size, lde.w);
  std::copy(lde.v.begin(), lde.v.begin() + size * lde.w, head.v.begin());
  return bit_reverse_rows(head);
}
}  // namespace

// ------------------------------------------------------------------ prove (src/prover.rs:290-603)
Proof prove(const System& sys, const std::vector<std::vector<u64>>& claims, Witness&& witness, StageTimes* times) {
  double t_begin = now_s(), t0;
  const Params& prm = sys.params;
  unsigned lb = (unsigned)prm.log_blowup;
  size_t C = sys.circuits.size();
  if (witness.traces.size() != C || witness.lookups.size() != C) throw std::runtime_error("witness/circuit count mismatch");
  Challenger ch = sys.new_challenger();
  sys.observe_shape(ch);
  Proof proof;
  std::vector<size_t> active_idx;
  std::vector<int> active_pos(C, -1);
  for (size_t i = 0; i < C; i++) {
    bool a = witness.traces[i].h > 0;
    proof.active.push_back(a);
    ch.observe(a ? 1 : 0);
    if (a) {
      active_pos[i] = (int)active_idx.size();
      active_idx.push_back(i);
    }
  }
  if (active_idx.empty()) throw std::runtime_error("cannot prove with every circuit deactivated");

  // stage 1 commit
  t0 = now_s();
  std::vector<unsigned> log_degrees;
  MerkleTree s1_tree;
  {
    std::vector<Mat> ldes;
    for (size_t ci : active_idx) {
      const Mat& tr = witness.traces[ci];
      if (tr.h & (tr.h - 1)) throw std::runtime_error("trace height must be a power of two");
      if (tr.w != sys.circuits[ci].main_width) throw std::runtime_error("trace width mismatch");
      log_degrees.push_back(log2_strict(tr.h));
      ldes.push_back(coset_lde_bitrev(tr, lb, F_GENERATOR));
    }
    mmcs_commit(std::move(ldes), (unsigned)prm.cap_height, s1_tree);
  }
  proof.stage1_commit = s1_tree.cap();
  if (times) times->stage1_commit = now_s() - t0;

  if (sys.has_pre) ch.observe_cap(sys.pre_commit);
  ch.observe_cap(proof.stage1_commit);
  for (unsigned ld : log_degrees) ch.observe(ld);
  ch.observe(f_from_u64((u64)claims.size()));
  for (auto& c : claims) {
    ch.observe(f_from_u64((u64)c.size()));
    for (u64 x : c) ch.observe(x);
  }
  EF beta = ch.sample_ext();
  ch.observe_ext(beta);
  EF gamma = ch.sample_ext();
  ch.observe_ext(gamma);
  EF acc = claims_accumulator(claims, beta, gamma);

  // lookup construction
  t0 = now_s();
  std::vector<std::vector<EF>> s2_traces;
  {
    std::vector<LookupValues> active_lookups;
    for (size_t ci : active_idx) {
      LookupValues& lv = witness.lookups[ci];
      if (lv.height != witness.traces[ci].h || lv.num_lookups != sys.circuits[ci].num_lookups)
        throw std::runtime_error("lookup values shape mismatch");
      active_lookups.push_back(std::move(lv));
    }
    stage_2_traces(active_lookups, beta, gamma, acc, s2_traces, proof.intermediate_accumulators);
  }
  if (times) times->lookup_construction = now_s() - t0;

  // stage 2 commit
  t0 = now_s();
  MerkleTree s2_tree;
  {
    std::vector<Mat> ldes;
    for (size_t pos = 0; pos < active_idx.size(); pos++) {
      const Circuit& c = sys.circuits[active_idx[pos]];
      size_t n = size_t(1) << log_degrees[pos];
      Mat flat(n, c.stage2_width);
      const std::vector<EF>& t = s2_traces[pos];
      for (size_t i = 0; i < t.size(); i++) ef_to(t[i], &flat.v[EXT_D * i]);
      ldes.push_back(coset_lde_bitrev(flat, lb, F_GENERATOR));
    }
    s2_traces.clear();
    mmcs_commit(std::move(ldes), (unsigned)prm.cap_height, s2_tree);
  }
  proof.stage2_commit = s2_tree.cap();
  if (times) times->stage2_commit = now_s() - t0;
  ch.observe_cap(proof.stage2_commit);
  for (auto& a : proof.intermediate_accumulators) ch.observe_ext(a);
  EF alpha = ch.sample_ext();

  // quotient
  t0 = now_s();
  MerkleTree q_tree;
  {
    std::vector<Mat> q_ldes;
    for (size_t pos = 0; pos < active_idx.size(); pos++) {
      size_t ci = active_idx[pos];
      const Circuit& c = sys.circuits[ci];
      size_t qd = c.quotient_degree();
      unsigned log_q = log2_strict(qd), log_n = log_degrees[pos];
      size_t qsize = size_t(1) << (log_n + log_q);
      EF next_acc = proof.intermediate_accumulators[pos];
      Mat pre_q;
      bool has_pre = sys.has_pre && sys.pre_indices[ci] >= 0;
      if (has_pre) pre_q = evaluations_on_domain(sys.pre_tree.mats[sys.pre_indices[ci]], qsize);
      Mat s1_q = evaluations_on_domain(s1_tree.mats[pos], qsize);
      Mat s2_q = evaluations_on_domain(s2_tree.mats[pos], qsize);
      u64 publics[4 * EXT_D];  // src/lookup.rs:82-84: beta, gamma, acc_initial, acc_final
      ef_to(beta, publics), ef_to(gamma, publics + EXT_D), ef_to(acc, publics + 2 * EXT_D), ef_to(next_acc, publics + 3 * EXT_D);
      std::vector<EF> qv = quotient_values(c, publics, log_n, log_q, has_pre ? &pre_q : nullptr, s1_q, s2_q, alpha);
      Mat qflat(qsize, EXT_D);
      for (size_t i = 0; i < qsize; i++) ef_to(qv[i], &qflat.v[EXT_D * i]);
      acc = next_acc;
      Mat sliced = shifted_quotient_slices(qflat, F_GENERATOR, qd);
      q_ldes.push_back(lde_from_shifted_coefficients(sliced, lb));
    }
    mmcs_commit(std::move(q_ldes), (unsigned)prm.cap_height, q_tree);
  }
  proof.quotient_commit = q_tree.cap();
  ch.observe_cap(proof.quotient_commit);
  if (times) times->quotient = now_s() - t0;

  // opening
  t0 = now_s();
  EF zeta = ch.sample_ext();
  std::vector<OpenRound> rounds(3);
  rounds[0].tree = &s1_tree;
  rounds[1].tree = &s2_tree;
  rounds[2].tree = &q_tree;
  for (unsigned ld : log_degrees) {
    EF zn = ef_mul_base(zeta, f_two_adic_generator(ld));
    rounds[0].points.push_back({zeta, zn});
    rounds[1].points.push_back({zeta, zn});
    rounds[2].points.push_back({zeta});
  }
  if (sys.has_pre) {
    OpenRound r0;
    r0.tree = &sys.pre_tree;
    for (size_t ci = 0; ci < C; ci++) {
      if (sys.pre_indices[ci] < 0) continue;
      if (active_pos[ci] >= 0) {
        EF zn = ef_mul_base(zeta, f_two_adic_generator(log_degrees[active_pos[ci]]));
        r0.points.push_back({zeta, zn});
      } else {
        r0.points.push_back({});
      }
    }
    rounds.push_back(std::move(r0));
  }
  std::vector<OpenedRound> opened;
  pcs_open(prm, rounds, ch, opened, proof.opening_proof);
  proof.stage1_opened = std::move(opened[0]);
  proof.stage2_opened = std::move(opened[1]);
  proof.quotient_opened = std::move(opened[2]);
  if (sys.has_pre) {
    proof.has_pre_opened = true;
    proof.pre_opened = std::move(opened[3]);
  }
  for (unsigned ld : log_degrees) proof.log_degrees.push_back((uint8_t)ld);
  if (times) {
    times->fri_open = now_s() - t0;
    times->total = now_s() - t_begin;
  }
  return proof;
}

// ------------------------------------------------------------------ verify
namespace {
struct RoundClaim {
  std::vector<Digest> commit;
  // per matrix: log trace height, and (point, values) pairs
  std::vector<unsigned> log_n;
  std::vector<std::vector<std::pair<EF, std::vector<EF>>>> mats;
};

// [UPSTREAM-RECALL p3-fri TwoAdicFriPcs::verify + verifier::verify_fri / verify_query]
bool pcs_verify(const Params& prm, const std::vector<RoundClaim>& rounds, const FriProof& proof, Challenger& ch) {
  unsigned lb = (unsigned)prm.log_blowup;
  for (auto& r : rounds)
    for (auto& m : r.mats)
      for (auto& pv : m)
        for (auto& y : pv.second) ch.observe_ext(y);
  EF alpha = ch.sample_ext();
  size_t nrounds = proof.commit_phase_commits.size();
  if (proof.commit_pow_witnesses.size() != nrounds) return false;
  // the rounds' arities are read off the first query's openings (every query must repeat them, and each is checked against
  // the schedule once the input heights are known below); their sum places the tallest input
  std::vector<unsigned> arities(nrounds, 1);
  if (!proof.query_proofs.empty()) {
    const QueryProof& q0 = proof.query_proofs[0];
    if (q0.commit_phase_openings.size() != nrounds) return false;
    for (size_t i = 0; i < nrounds; i++) {
      arities[i] = q0.commit_phase_openings[i].log_arity;
      if (arities[i] < 1 || arities[i] > prm.max_log_arity) return false;
    }
  }
  unsigned log_gmax = (unsigned)(lb + prm.log_final_poly_len);
  for (unsigned a : arities) log_gmax += a;
  if (log_gmax > 62) return false;
  std::vector<EF> betas;
  for (size_t i = 0; i < nrounds; i++) {
    ch.observe_cap(proof.commit_phase_commits[i]);
    if (!ch.check_witness((unsigned)prm.commit_pow_bits, proof.commit_pow_witnesses[i])) return false;
    betas.push_back(ch.sample_ext());
  }
  if (proof.final_poly.size() != (size_t(1) << prm.log_final_poly_len)) return false;
  for (auto& c : proof.final_poly) ch.observe_ext(c);
  if (proof.query_proofs.size() != prm.num_queries) return false;
  if (!ch.check_witness((unsigned)prm.query_pow_bits, proof.query_pow_witness)) return false;
  unsigned log_final_height = (unsigned)(lb + prm.log_final_poly_len);
  for (auto& qp : proof.query_proofs) {
    size_t index = ch.sample_bits(log_gmax);
    if (qp.input_proof.size() != rounds.size()) return false;
    // open_input: per-height reduced openings
    std::map<unsigned, std::pair<EF, EF>> ro;  // log_height -> (alpha_pow, reduced opening)
    for (size_t ri = 0; ri < rounds.size(); ri++) {
      const RoundClaim& r = rounds[ri];
      const BatchOpening& bo = qp.input_proof[ri];
      if (bo.opened_values.size() != r.mats.size()) return false;
      std::vector<Dim> dims;
      unsigned log_bmax = 0;
      for (size_t mi = 0; mi < r.mats.size(); mi++) {
        size_t w = bo.opened_values[mi].size();
        dims.push_back(Dim{w, size_t(1) << (r.log_n[mi] + lb)});
        log_bmax = std::max(log_bmax, r.log_n[mi] + lb);
      }
      if (log_bmax > log_gmax) return false;
      if (!mmcs_verify_batch(r.commit, dims, index >> (log_gmax - log_bmax), bo)) return false;
      for (size_t mi = 0; mi < r.mats.size(); mi++) {
        unsigned lh = r.log_n[mi] + lb;
        size_t rev = bitrev(index >> (log_gmax - lh), lh);
        u64 x = f_mul(F_GENERATOR, f_pow(f_two_adic_generator(lh), rev));
        auto it = ro.find(lh);
        if (it == ro.end()) it = ro.emplace(lh, std::make_pair(ef(1), ef(0))).first;
        for (auto& pv : r.mats[mi]) {
          if (pv.second.size() != bo.opened_values[mi].size()) return false;
          EF quot = ef_inv(ef_sub(pv.first, ef(x)));
          for (size_t c = 0; c < pv.second.size(); c++) {
            EF diff = ef_sub(pv.second[c], ef(bo.opened_values[mi][c]));
            it->second.second = ef_add(it->second.second, ef_mul(ef_mul(it->second.first, diff), quot));
            it->second.first = ef_mul(it->second.first, alpha);
          }
        }
      }
    }
    // a height-1 trace gives a constant polynomial: its reduced opening must vanish
    auto low = ro.find(lb);
    if (low != ro.end() && log_final_height >= lb && lb < log_gmax) {
      if (!ef_is_zero(low->second.second)) return false;
      ro.erase(low);
    }
    // verify_query
    if (qp.commit_phase_openings.size() != nrounds) return false;
    auto it = ro.rbegin();
    if (it == ro.rend() || it->first != log_gmax) return false;
    EF folded = it->second.second;
    ++it;
    size_t idx = index;
    unsigned log_height = log_gmax;
    for (size_t i = 0; i < nrounds; i++) {
      const CommitPhaseStep& st = qp.commit_phase_openings[i];
      const unsigned la = arities[i];
      const size_t ar = size_t(1) << la;
      // the arity the prover had to choose here (compute_log_arity_for_round)
      if (log_height <= log_final_height) return false;
      if (la != fri_log_arity_for_round(log_height, it != ro.rend() ? (int)it->first : -1, log_final_height, (unsigned)prm.max_log_arity)) return false;
      if (st.log_arity != la || st.sibling_values.size() != ar - 1) return false;
      const unsigned log_folded_height = log_height - la;
      const size_t own = idx & (ar - 1), row = idx >> la;
      std::vector<EF> evals(ar);
      for (size_t j = 0, k = 0; j < ar; j++) evals[j] = j == own ? folded : st.sibling_values[k++];
      BatchOpening bo;
      bo.opened_values.emplace_back(ar * EXT_D);
      for (size_t j = 0; j < ar; j++) ef_to(evals[j], bo.opened_values[0].data() + EXT_D * j);
      bo.proof = st.proof;
      if (!mmcs_verify_batch(proof.commit_phase_commits[i], {Dim{ar * EXT_D, size_t(1) << log_folded_height}}, row, bo))
        return false;
      idx = row;
      // fold_row: the polynomial of degree < 2^a through the row's points, evaluated at beta - as binary steps with
      // beta, beta^2, ...: a step interpolates (x0, e0), (-x0, e1) and evaluates at the step's challenge
      EF bp = betas[i];
      for (unsigned s = 0; s < la; s++) {
        const unsigned lf = log_height - s - 1;  // log height of this step's output
        const size_t m = ar >> (s + 1);
        for (size_t t = 0; t < m; t++) {
          u64 x0 = f_pow(f_two_adic_generator(lf + 1), bitrev(idx * m + t, lf));
          u64 x1 = f_neg(x0);
          EF slope = ef_mul_base(ef_sub(evals[2 * t + 1], evals[2 * t]), f_inv(f_sub(x1, x0)));
          evals[t] = ef_add(evals[2 * t], ef_mul(ef_sub(bp, ef(x0)), slope));
        }
        bp = ef_square(bp);
      }
      folded = evals[0];
      log_height = log_folded_height;
      if (it != ro.rend() && it->first == log_folded_height) {
        folded = ef_add(folded, ef_mul(fri_roll_in_factor(betas[i], la), it->second.second));
        ++it;
      }
    }
    if (it != ro.rend()) return false;
    // final polynomial at x = w_{gmax}^{bitrev(idx, log_gmax)} (idx is now log_final_height bits wide)
    u64 x = f_pow(f_two_adic_generator(log_gmax), bitrev(idx, log_gmax));
    EF eval = ef(0);
    for (size_t k = proof.final_poly.size(); k-- > 0;) eval = ef_add(ef_mul_base(eval, x), proof.final_poly[k]);
    if (!ef_eq(eval, folded)) return false;
  }
  return true;
}
}  // namespace

void pcs_open_rounds(const Params& prm, const std::vector<PcsOpenRound>& rounds, Challenger& ch, std::vector<OpenedRound>& opened,
                     FriProof& proof) {
  std::vector<OpenRound> rs;
  for (auto& r : rounds) rs.push_back(OpenRound{r.tree, r.points});
  pcs_open(prm, rs, ch, opened, proof);
}
bool pcs_verify_rounds(const Params& prm, const std::vector<PcsVerifyRound>& rounds, const FriProof& proof, Challenger& ch) {
  std::vector<RoundClaim> rs;
  for (auto& r : rounds) {
    RoundClaim c;
    c.commit = r.commit, c.log_n = r.log_n, c.mats = r.mats;
    rs.push_back(std::move(c));
  }
  return pcs_verify(prm, rs, proof, ch);
}

VerifyError verify(const System& sys, const std::vector<std::vector<u64>>& claims, const Proof& proof) {
  const Params& prm = sys.params;
  size_t C = sys.circuits.size();
  // ---- verify_shape, src/verifier.rs:536-695
  if (C == 0) return V_INVALID_SYSTEM;
  if (proof.active.size() != C) return V_INVALID_SHAPE;
  std::vector<size_t> active_idx;
  std::vector<int> active_pos(C, -1);
  for (size_t i = 0; i < C; i++)
    if (proof.active[i]) {
      active_pos[i] = (int)active_idx.size();
      active_idx.push_back(i);
    }
  size_t na = active_idx.size();
  if (na == 0) return V_INVALID_SHAPE;
  if (proof.log_degrees.size() != na) return V_INVALID_SHAPE;
  size_t num_pre = 0;
  for (int pi : sys.pre_indices) num_pre += pi >= 0;
  if (sys.has_pre != (num_pre != 0)) return V_INVALID_SYSTEM;
  if ((proof.has_pre_opened ? proof.pre_opened.size() : 0) != num_pre) return V_INVALID_SHAPE;
  for (size_t ci = 0; ci < C; ci++)
    if (sys.pre_indices[ci] >= 0 && !proof.active[ci] && proof.pre_opened[sys.pre_indices[ci]].size() != 0)
      return V_INVALID_SHAPE;
  if (proof.stage1_opened.size() != na || proof.stage2_opened.size() != na) return V_INVALID_SHAPE;
  for (size_t pos = 0; pos < na; pos++) {
    size_t ci = active_idx[pos];
    const Circuit& c = sys.circuits[ci];
    int slot = sys.pre_indices[ci];
    if (proof.stage1_opened[pos].size() != 2 || proof.stage2_opened[pos].size() != 2) return V_INVALID_SHAPE;
    if (slot >= 0 && proof.pre_opened[slot].size() != 2) return V_INVALID_SHAPE;
    for (int j = 0; j < 2; j++) {
      if (slot >= 0 && proof.pre_opened[slot][j].size() != c.pre_width) return V_INVALID_SHAPE;
      if (proof.stage1_opened[pos][j].size() != c.main_width) return V_INVALID_SHAPE;
      if (proof.stage2_opened[pos][j].size() != c.stage2_width) return V_INVALID_SHAPE;
    }
  }
  std::vector<size_t> qdeg;
  size_t max_log_degree = F_TWO_ADICITY - prm.log_blowup;  // src/types.rs:131, baby_bear_config.rs:87
  for (size_t pos = 0; pos < na; pos++) {
    size_t qd = sys.circuits[active_idx[pos]].quotient_degree();
    if (proof.log_degrees[pos] + log2_strict(qd) > max_log_degree) return V_INVALID_SHAPE;
    qdeg.push_back(qd);
  }
  if (proof.quotient_opened.size() != na) return V_INVALID_SHAPE;
  for (size_t pos = 0; pos < na; pos++) {
    if (proof.quotient_opened[pos].size() != 1) return V_INVALID_SHAPE;
    if (proof.quotient_opened[pos][0].size() != qdeg[pos] * EXT_D) return V_INVALID_SHAPE;
  }
  if (proof.intermediate_accumulators.size() != na) return V_INVALID_SHAPE;

  // ---- src/verifier.rs:242-246
  {
    if (!ef_is_zero(proof.intermediate_accumulators.back())) return V_UNBALANCED;
  }
  // ---- transcript replay, src/verifier.rs:255-326
  Challenger ch = sys.new_challenger();
  sys.observe_shape(ch);
  for (auto a : proof.active) ch.observe(a ? 1 : 0);
  if (sys.has_pre) ch.observe_cap(sys.pre_commit);
  ch.observe_cap(proof.stage1_commit);
  for (auto ld : proof.log_degrees) ch.observe((u64)ld);
  ch.observe(f_from_u64((u64)claims.size()));
  for (auto& c : claims) {
    ch.observe(f_from_u64((u64)c.size()));
    for (u64 x : c) ch.observe(x);
  }
  EF beta = ch.sample_ext();
  ch.observe_ext(beta);
  EF gamma = ch.sample_ext();
  ch.observe_ext(gamma);
  ch.observe_cap(proof.stage2_commit);
  for (auto& a : proof.intermediate_accumulators) ch.observe_ext(a);
  EF acc = claims_accumulator(claims, beta, gamma);
  EF alpha = ch.sample_ext();
  ch.observe_cap(proof.quotient_commit);
  EF zeta = ch.sample_ext();

  std::vector<RoundClaim> rounds(3);
  rounds[0].commit = proof.stage1_commit;
  rounds[1].commit = proof.stage2_commit;
  rounds[2].commit = proof.quotient_commit;
  for (size_t pos = 0; pos < na; pos++) {
    unsigned ld = proof.log_degrees[pos];
    EF zn = ef_mul_base(zeta, f_two_adic_generator(ld));
    rounds[0].log_n.push_back(ld);
    rounds[0].mats.push_back({{zeta, proof.stage1_opened[pos][0]}, {zn, proof.stage1_opened[pos][1]}});
    rounds[1].log_n.push_back(ld);
    rounds[1].mats.push_back({{zeta, proof.stage2_opened[pos][0]}, {zn, proof.stage2_opened[pos][1]}});
    rounds[2].log_n.push_back(ld);
    rounds[2].mats.push_back({{zeta, proof.quotient_opened[pos][0]}});
  }
  if (sys.has_pre) {
    RoundClaim r0;
    r0.commit = sys.pre_commit;
    for (size_t ci = 0; ci < C; ci++) {
      int slot = sys.pre_indices[ci];
      if (slot < 0) continue;
      if (active_pos[ci] >= 0) {
        unsigned ld = proof.log_degrees[active_pos[ci]];
        EF zn = ef_mul_base(zeta, f_two_adic_generator(ld));
        r0.log_n.push_back(ld);
        r0.mats.push_back({{zeta, proof.pre_opened[slot][0]}, {zn, proof.pre_opened[slot][1]}});
      } else {
        r0.log_n.push_back(log2_strict(sys.circuits[ci].pre_height));
        r0.mats.push_back({});
      }
    }
    rounds.push_back(std::move(r0));
  }
  if (!pcs_verify(prm, rounds, proof.opening_proof, ch)) return V_INVALID_OPENING;

  // ---- OOD check per circuit, src/verifier.rs:419-530
  for (size_t pos = 0; pos < na; pos++) {
    size_t ci = active_idx[pos];
    const Circuit& c = sys.circuits[ci];
    unsigned ld = proof.log_degrees[pos];
    EF next_acc = proof.intermediate_accumulators[pos];
    // selectors_at_point [UPSTREAM-RECALL]: z_h = zeta^n - 1
    u64 g_inv = f_inv(f_two_adic_generator(ld));
    EF zh = ef_sub(ef_exp_pow2(zeta, ld), ef(1));
    EF is_first = ef_mul(zh, ef_inv(ef_sub(zeta, ef(1))));
    EF is_last = ef_mul(zh, ef_inv(ef_sub(zeta, ef(g_inv))));
    EF is_trans = ef_sub(zeta, ef(g_inv));
    EF inv_van = ef_inv(zh);
    u64 n_val = (u64(1) << ld) % F_P;
    u64 inj_norm = f_inv(f_mul(n_val, f_two_adic_generator(ld)));
    EF publics[4 * EXT_D];
    const EF four[4] = {beta, gamma, acc, next_acc};
    for (int k = 0; k < 4; k++)
      for (unsigned d = 0; d < EXT_D; d++) publics[EXT_D * k + d] = ef(four[k].c[d]);
    int slot = sys.pre_indices[ci];
    View<ExtOps> v;
    v.pre[0] = slot >= 0 ? proof.pre_opened[slot][0].data() : nullptr;
    v.pre[1] = slot >= 0 ? proof.pre_opened[slot][1].data() : nullptr;
    v.main[0] = proof.stage1_opened[pos][0].data();
    v.main[1] = proof.stage1_opened[pos][1].data();
    v.s2[0] = proof.stage2_opened[pos][0].data();
    v.s2[1] = proof.stage2_opened[pos][1].data();
    v.publics = publics;
    v.is_first = is_first;
    v.is_last = is_last;
    v.is_trans = is_trans;
    std::vector<EF> buf, cv;
    sweep<ExtOps>(c, v, buf, c.nodes.size());
    for (auto z : c.zeros) cv.push_back(buf[z]);
    EF delta_scaled[EXT_D];
    for (unsigned d = 0; d < EXT_D; d++)
      delta_scaled[d] = ef_mul_base(ef_sub(publics[3 * EXT_D + d], publics[2 * EXT_D + d]), inj_norm);
    logup_constraint_values<ExtOps>(c, buf, v.s2[0], v.s2[1], publics, delta_scaled, is_last, cv);
    if (cv.size() != c.constraint_count) return V_INVALID_SYSTEM;
    EF comp = ef(0);
    for (auto& x : cv) comp = ef_add(ef_mul(comp, alpha), x);
    // Q(zeta) = sum_i zeta^{i n} c_i(zeta), c_i = sum_k coord_k * X^k
    const std::vector<EF>& qrow = proof.quotient_opened[pos][0];
    EF zpn = ef_exp_pow2(zeta, ld), zp = ef(1), quot = ef(0);
    for (size_t i = 0; i < qdeg[pos]; i++) {
      EF chunk = ef(0);
      for (unsigned d = 0; d < EXT_D; d++) chunk = ef_add(chunk, ef_mul(qrow[EXT_D * i + d], ef_basis(d)));
      quot = ef_add(quot, ef_mul(zp, chunk));
      zp = ef_mul(zp, zpn);
    }
    if (!ef_eq(ef_mul(comp, inv_van), quot)) return V_OOD_MISMATCH;
    acc = next_acc;
  }
  return V_OK;
}

}  // namespace mso
