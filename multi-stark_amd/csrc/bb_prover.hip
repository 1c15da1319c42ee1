// Host side of the BabyBear / Poseidon2 path: System / witness mirrors, the DuplexChallenger transcript and
// System::prove_multiple_claims (/root/reference/src/prover.rs:290-603) instantiated for the reference's second
// configuration (src/test_circuits/baby_bear_config.rs:28-127). The transcript runs on the host (a few hundred
// Poseidon2 permutations per proof); every heavy step is a launch into bb_kernels.hip. The C ABI is include/mstark_bb.h.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>

#include "../../include/mstark_bb.h"
#include "bb.h"

namespace msbb {

using msamd::PNode;

static unsigned log2_strict(size_t n) {
  unsigned l = 0;
  while ((size_t(1) << l) < n) l++;
  return l;
}
static size_t bitrev_host(size_t x, unsigned bits) {
  size_t r = 0;
  for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
  return r;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// DuplexChallenger<BabyBear, Poseidon2BabyBear<16>, 16, 8> (baby_bear_config.rs:37), values in Montgomery form
struct Challenger {
  const Poseidon2* perm;
  u32 state[16];
  std::vector<u32> input, output;
  explicit Challenger(const Poseidon2* p) : perm(p) {
    for (auto& x : state) x = 0;
  }
  void duplexing() {
    for (size_t i = 0; i < input.size(); i++) state[i] = input[i];
    input.clear();
    bb_poseidon2(*perm, state);
    output.assign(state, state + 8);
  }
  void observe(u32 m) {
    output.clear();
    input.push_back(m);
    if (input.size() == 8) duplexing();
  }
  void observe_usize(u64 x) { observe(bb_to_monty((u32)(x % BB_P))); }  // Val::from_usize
  void observe_e4(E4 e) {
    for (int k = 0; k < 4; k++) observe(e.c[k]);
  }
  void observe_cap(const std::vector<Digest8>& cap) {
    for (auto& d : cap)
      for (int k = 0; k < 8; k++) observe(d.w[k]);
  }
  u32 sample() {
    if (!input.empty() || output.empty()) duplexing();
    u32 v = output.back();
    output.pop_back();
    return v;
  }
  E4 sample_e4() {
    E4 e;
    for (int k = 0; k < 4; k++) e.c[k] = sample();
    return e;
  }
  size_t sample_bits(unsigned bits) { return (size_t)(bb_from_monty(sample()) & ((1u << bits) - 1)); }
  // smallest witness (canonical value); ZERO at 0 bits - the deterministic rule of src/types.rs:72-81
  u32 grind(unsigned bits) {
    if (bits == 0) return 0;
    for (u32 w = 0; w < BB_P; w++) {
      Challenger c = *this;
      c.observe(bb_to_monty(w));
      if (c.sample_bits(bits) == 0) {
        observe(bb_to_monty(w));
        sample_bits(bits);
        return w;
      }
    }
    throw std::runtime_error("grind: no witness");
  }
};

struct Params {
  u64 log_blowup = 1, cap_height = 0, log_final_poly_len = 0, max_log_arity = 1, num_queries = 1, commit_pow_bits = 0, query_pow_bits = 0;
};
struct BCircuit {
  std::vector<PNode> nodes;
  std::vector<uint32_t> degrees, zeros;
  std::vector<std::pair<uint32_t, std::vector<uint32_t>>> lookups;
  size_t main_width = 0, pre_width = 0, pre_height = 0, num_lookups = 0, stage2_width = 0, constraint_count = 0, max_constraint_degree = 0,
         args_width = 0, lookup_prefix_len = 0;
  BProgram prog;
  msamd::JitKernel quotient_jit;  // this circuit's quotient kernel, compiled at system creation (quotient_jit.hip); may be empty
  BLookupsDev lk;
  DBuf<u32> d_zeros;
  BMat pre;  // preprocessed trace (column-major, Montgomery), for witness preparation
  size_t quotient_degree() const {
    size_t d = (max_constraint_degree > 2 ? max_constraint_degree : 2) - 1, q = 1;
    while (q < d) q <<= 1;
    return q;
  }
};
struct BSystem {
  Ctx* ctx = nullptr;
  Params params;
  Poseidon2 perm;
  DBuf<Poseidon2> d_perm;
  std::vector<BCircuit> circuits;
  bool has_pre = false;
  std::vector<Digest8> pre_commit;
  std::vector<int> pre_indices;
  BPcsData pre_data;
  std::vector<u32> seed;  // Montgomery form
};
struct BWitness {
  BSystem* sys = nullptr;
  std::vector<size_t> heights;
  std::vector<BMat> traces;
  std::vector<std::vector<u32>> claims;  // canonical (the transcript absorbs them on the host)
  DBuf<u32> d_claim_data;                // Montgomery form, concatenated
  DBuf<u64> d_claim_offs;
  // host-resident form (msbb_witness_create_host): nothing lives in HBM between proofs; every prove() uploads the caller's
  // (page-locked) trace buffers and the claims, and gives the device copies back when it is done
  bool host_resident = false;
  bool pinned = true;  // every trace buffer could be page-locked (otherwise the uploads go through the context's bounce buffer)
  std::vector<const u32*> h_traces;
  std::vector<void*> registered;
  std::vector<u32> h_claims_monty;
  std::vector<u64> h_claim_offs;
  ~BWitness() {
    if (!registered.empty() && sys && sys->ctx) {  // nothing may still be reading the caller's ranges when they lose their page lock
      (void)hipSetDevice(sys->ctx->device);
      (void)hipStreamSynchronize(sys->ctx->main_stream);
    }
    for (void* p : registered) msamd::host_range_unpin(p);
  }
};

static std::vector<Digest8> tree_cap(Ctx& ctx, const BTree& t) {
  size_t l = t.cap_layer();
  std::vector<Digest8> cap(t.sizes[l]);
  ctx.d2h(cap.data(), t.layers[l].p, cap.size() * sizeof(Digest8));
  return cap;
}

namespace {
struct Reader {
  const uint8_t* p;
  size_t n, off = 0;
  u64 word() {
    if (off + 8 > n) throw std::runtime_error("system blob truncated");
    u64 v = 0;
    for (int k = 0; k < 8; k++) v |= (u64)p[off + k] << (8 * k);
    off += 8;
    return v;
  }
};
}  // namespace
static const u64 BLOB_MAGIC = 0x31304259534D0000ULL;  // "\0\0MSYB01"

static void set_internal_diag(Poseidon2& k) {
  auto m = [](u32 canonical) { return bb_to_monty(canonical); };
  u32 half = bb_inv(m(2)), i8 = bb_inv(m(256)), i27 = bb_inv(m(1u << 27));
  u32 t[16] = {bb_neg(m(2)), m(1), m(2), half, m(3), m(4), bb_neg(half), bb_neg(m(3)), bb_neg(m(4)), i8, bb_inv(m(4)), bb_inv(m(8)), i27,
               bb_neg(i8), bb_neg(bb_inv(m(16))), bb_neg(i27)};
  for (int i = 0; i < 16; i++) k.diag[i] = t[i];
}

std::unique_ptr<BSystem> system_from_blob(Ctx& ctx, const uint8_t* blob, size_t len) {
  HIP_CHECK(hipSetDevice(ctx.device));
  Reader rd{blob, len};
  if (rd.word() != BLOB_MAGIC) throw std::runtime_error("bad system blob magic (expected a BabyBear / Poseidon2 system)");
  std::unique_ptr<BSystem> sys(new BSystem());
  sys->ctx = &ctx;
  Params& p = sys->params;
  p.log_blowup = rd.word(), p.cap_height = rd.word(), p.log_final_poly_len = rd.word(), p.max_log_arity = rd.word();
  p.num_queries = rd.word(), p.commit_pow_bits = rd.word(), p.query_pow_bits = rd.word();
  if (p.max_log_arity < 1 || p.max_log_arity > BB_FRI_MAX_LOG_ARITY) throw std::runtime_error("max_log_arity must be 1 .. 6");
  if (p.log_blowup < 1 || p.log_blowup > 8) throw std::runtime_error("log_blowup out of range");
  if (p.commit_pow_bits > 24 || p.query_pow_bits > 24) throw std::runtime_error("proof-of-work bits out of range");
  for (int r = 0; r < 8; r++)
    for (int i = 0; i < 16; i++) {
      u64 v = rd.word();
      if (v >= BB_P) throw std::runtime_error("non-canonical round constant");
      sys->perm.external[r][i] = bb_to_monty((u32)v);
    }
  for (int r = 0; r < 13; r++) {
    u64 v = rd.word();
    if (v >= BB_P) throw std::runtime_error("non-canonical round constant");
    sys->perm.internal[r] = bb_to_monty((u32)v);
  }
  set_internal_diag(sys->perm);
  sys->d_perm = DBuf<Poseidon2>(ctx, 1);
  ctx.h2d(sys->d_perm.p, &sys->perm, sizeof(Poseidon2));
  {
    const char* tag = "multi-stark/v0";  // baby_bear_config.rs:72-86
    for (int i = 0; i < 14; i++) sys->seed.push_back(bb_to_monty((u32)(uint8_t)tag[i]));
    const u64 ps[7] = {p.log_blowup, p.cap_height, p.log_final_poly_len, p.max_log_arity, p.num_queries, p.commit_pow_bits, p.query_pow_bits};
    for (u64 x : ps) sys->seed.push_back(bb_to_monty((u32)(x % BB_P)));
  }
  const size_t D = 4;
  size_t nc = rd.word();
  std::vector<BMat> pre_ldes;
  for (size_t ci = 0; ci < nc; ci++) {
    sys->circuits.emplace_back();
    BCircuit& c = sys->circuits.back();
    c.main_width = rd.word(), c.pre_width = rd.word(), c.pre_height = rd.word();
    size_t nn = rd.word(), nz = rd.word(), nl = rd.word();
    c.num_lookups = nl;
    c.stage2_width = std::max<size_t>(nl, 1) * D;  // src/lookup.rs:90-92
    c.nodes.resize(nn);
    c.degrees.resize(nn);
    for (size_t i = 0; i < nn; i++) {
      u64 w0 = rd.word();
      PNode& nd = c.nodes[i];
      nd.kind = (uint32_t)(w0 & 0xff), nd.source = (uint32_t)((w0 >> 8) & 0xff), nd.offset = (uint32_t)((w0 >> 16) & 0xff);
      nd.a = rd.word(), nd.b = rd.word();
      auto child = [&](u64 id) -> uint32_t {
        if (id >= i) throw std::runtime_error("node program is not topologically ordered");
        return c.degrees[id];
      };
      uint32_t deg = 0;
      switch (nd.kind) {  // src/graph.rs:242-252
        case msamd::OP_CONST:
          if (nd.a >= BB_P) throw std::runtime_error("non-canonical constant in node program");
          break;
        case msamd::OP_PUBLIC:
          if (nd.a >= 4 * D) throw std::runtime_error("public index out of range");
          break;
        case msamd::OP_IS_TRANS: break;
        case msamd::OP_VAR: {
          size_t width = nd.source == 0 ? c.pre_width : nd.source == 1 ? c.main_width : c.stage2_width;
          if (nd.source > 2 || nd.offset > 1 || nd.a >= width) throw std::runtime_error("column reference out of range");
          deg = 1;
          break;
        }
        case msamd::OP_IS_FIRST:
        case msamd::OP_IS_LAST: deg = 1; break;
        case msamd::OP_ADD:
        case msamd::OP_SUB: deg = std::max(child(nd.a), child(nd.b)); break;
        case msamd::OP_MUL: deg = child(nd.a) + child(nd.b); break;
        case msamd::OP_NEG: deg = child(nd.a); break;
        default: throw std::runtime_error("bad node kind");
      }
      c.degrees[i] = deg;
    }
    uint32_t graph_deg = 0;
    for (size_t i = 0; i < nz; i++) {
      u64 z = rd.word();
      if (z >= nn) throw std::runtime_error("constraint root out of range");
      c.zeros.push_back((uint32_t)z);
      graph_deg = std::max(graph_deg, c.degrees[z]);
    }
    uint32_t logup_deg = nl ? 0 : 1;  // src/lookup.rs:262-278
    std::vector<u32> lk_mult, lk_off(1, 0), lk_args;
    for (size_t j = 0; j < nl; j++) {
      u64 m = rd.word();
      if (m >= nn) throw std::runtime_error("lookup node out of range");
      size_t na = rd.word();
      std::vector<uint32_t> args;
      uint32_t msg = 0;
      c.lookup_prefix_len = std::max<size_t>(c.lookup_prefix_len, m + 1);
      for (size_t k = 0; k < na; k++) {
        u64 a = rd.word();
        if (a >= nn) throw std::runtime_error("lookup node out of range");
        args.push_back((uint32_t)a);
        lk_args.push_back((u32)a);
        msg = std::max(msg, c.degrees[a]);
        c.lookup_prefix_len = std::max<size_t>(c.lookup_prefix_len, a + 1);
      }
      lk_mult.push_back((u32)m);
      lk_off.push_back((u32)lk_args.size());
      c.args_width += na;
      logup_deg = std::max(logup_deg, std::max(msg + 1, c.degrees[m]));
      c.lookups.emplace_back((uint32_t)m, std::move(args));
    }
    for (size_t i = 0; i < c.lookup_prefix_len; i++)  // src/graph.rs: the lookup prefix lives in the base context
      if (c.nodes[i].kind == msamd::OP_PUBLIC || (c.nodes[i].kind == msamd::OP_VAR && c.nodes[i].source == 2))
        throw std::runtime_error("lookup expressions may only read trace columns and row selectors");
    c.constraint_count = nz + std::max<size_t>(nl, 1) * D;  // src/system.rs:151
    c.max_constraint_degree = std::max(graph_deg, logup_deg);
    if (c.quotient_degree() > (size_t(1) << p.log_blowup))  // src/system.rs:171-178
      throw std::runtime_error("circuit " + std::to_string(ci) + ": constraint degree needs a quotient degree beyond the blowup");
    bb_build_program(ctx, c.nodes, c.prog);
    msamd::bb_quotient_jit_build(c.nodes, c.zeros, c.lookups, c.quotient_jit);
    c.lk.L = nl;
    c.lk.mult = DBuf<u32>(ctx, std::max<size_t>(nl, 1));
    c.lk.arg_off = DBuf<u32>(ctx, nl + 1);
    c.lk.args = DBuf<u32>(ctx, std::max<size_t>(lk_args.size(), 1));
    c.d_zeros = DBuf<u32>(ctx, std::max<size_t>(nz, 1));
    if (nl) ctx.h2d(c.lk.mult.p, lk_mult.data(), nl * 4);
    ctx.h2d(c.lk.arg_off.p, lk_off.data(), (nl + 1) * 4);
    if (!lk_args.empty()) ctx.h2d(c.lk.args.p, lk_args.data(), lk_args.size() * 4);
    if (nz) ctx.h2d(c.d_zeros.p, c.zeros.data(), nz * 4);
    ctx.sync();
    if (c.pre_width) {
      if (c.pre_height == 0 || (c.pre_height & (c.pre_height - 1))) throw std::runtime_error("preprocessed height must be a power of two");
      if (log2_strict(c.pre_height) + p.log_blowup > BB_TWO_ADICITY) throw std::runtime_error("preprocessed trace too tall");
      std::vector<u32> rows(c.pre_height * c.pre_width);
      for (auto& x : rows) {
        u64 v = rd.word();
        if (v >= BB_P) throw std::runtime_error("non-canonical preprocessed value");
        x = (u32)v;
      }
      bb_upload_rows(ctx, rows.data(), c.pre_height, c.pre_width, c.pre);
      sys->pre_indices.push_back((int)pre_ldes.size());
      pre_ldes.emplace_back();
      bb_coset_lde(ctx, c.pre, (unsigned)p.log_blowup, pre_ldes.back());
    } else {
      c.pre_height = 0;
      sys->pre_indices.push_back(-1);
    }
  }
  if (rd.off != len) throw std::runtime_error("trailing bytes in system blob");
  if (!pre_ldes.empty()) {
    sys->has_pre = true;
    bb_commit(ctx, sys->d_perm.p, std::move(pre_ldes), (unsigned)p.cap_height, sys->pre_data);
    sys->pre_commit = tree_cap(ctx, sys->pre_data.tree);
  }
  ctx.sync();
  return sys;
}

std::unique_ptr<BWitness> witness_create(BSystem& sys, const u32* const* traces, const u64* heights, size_t n_claims, const u64* claim_offsets,
                                         const u32* claim_data) {
  Ctx& ctx = *sys.ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  std::unique_ptr<BWitness> w(new BWitness());
  w->sys = &sys;
  size_t C = sys.circuits.size();
  for (size_t ci = 0; ci < C; ci++) {
    const BCircuit& c = sys.circuits[ci];
    size_t h = (size_t)heights[ci];
    w->heights.push_back(h);
    w->traces.emplace_back();
    if (h == 0) continue;
    if (h & (h - 1)) throw std::runtime_error("trace height must be a power of two");
    if (log2_strict(h) + log2_strict(c.quotient_degree()) + sys.params.log_blowup > BB_TWO_ADICITY)  // baby_bear_config.rs:87
      throw std::runtime_error("trace too tall for the two-adicity of BabyBear");
    if (c.pre_width && h != c.pre_height) throw std::runtime_error("main trace height must equal preprocessed trace height");
    if (!traces[ci]) throw std::runtime_error("missing trace");
    for (size_t i = 0; i < h * c.main_width; i++)
      if (traces[ci][i] >= BB_P) throw std::runtime_error("non-canonical trace value");
    bb_upload_rows(ctx, traces[ci], h, c.main_width, w->traces.back());
  }
  std::vector<u32> flat;
  for (size_t i = 0; i < n_claims; i++) {
    if (claim_offsets[i + 1] < claim_offsets[i]) throw std::runtime_error("claim offsets must not decrease");
    w->claims.emplace_back(claim_data + claim_offsets[i], claim_data + claim_offsets[i + 1]);
    for (u32 x : w->claims.back()) {
      if (x >= BB_P) throw std::runtime_error("non-canonical claim value");
      flat.push_back(bb_to_monty(x));
    }
  }
  if (n_claims) {
    std::vector<u64> offs(n_claims + 1);
    for (size_t i = 0; i <= n_claims; i++) offs[i] = claim_offsets[i] - claim_offsets[0];
    w->d_claim_offs = DBuf<u64>(ctx, n_claims + 1);
    w->d_claim_data = DBuf<u32>(ctx, std::max<size_t>(flat.size(), 1));
    ctx.h2d(w->d_claim_offs.p, offs.data(), offs.size() * 8);
    if (!flat.empty()) ctx.h2d(w->d_claim_data.p, flat.data(), flat.size() * 4);
    ctx.sync();
  }
  return w;
}

// A SystemWitness that stays in host memory, which is what the reference's prove() is handed (src/prover.rs:290-295): values
// validated and buffers page-locked here (setup), uploaded by every prove().
std::unique_ptr<BWitness> witness_create_host(BSystem& sys, const u32* const* traces, const u64* heights, size_t n_claims,
                                              const u64* claim_offsets, const u32* claim_data, bool* pinned) {
  Ctx& ctx = *sys.ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  std::unique_ptr<BWitness> w(new BWitness());
  w->sys = &sys;
  w->host_resident = true;
  bool all_pinned = true;
  const size_t C = sys.circuits.size();
  for (size_t ci = 0; ci < C; ci++) {
    const BCircuit& c = sys.circuits[ci];
    const size_t h = (size_t)heights[ci];
    w->heights.push_back(h);
    w->traces.emplace_back();
    w->h_traces.push_back(nullptr);
    if (h == 0) continue;
    if (h & (h - 1)) throw std::runtime_error("trace height must be a power of two");
    if (log2_strict(h) + log2_strict(c.quotient_degree()) + sys.params.log_blowup > BB_TWO_ADICITY)
      throw std::runtime_error("trace too tall for the two-adicity of BabyBear");
    if (c.pre_width && h != c.pre_height) throw std::runtime_error("main trace height must equal preprocessed trace height");
    if (!traces[ci]) throw std::runtime_error("missing trace");
    for (size_t i = 0; i < h * c.main_width; i++)
      if (traces[ci][i] >= BB_P) throw std::runtime_error("non-canonical trace value");
    w->h_traces[ci] = traces[ci];
    const int r = msamd::host_range_pin(traces[ci], h * c.main_width * 4);  // (counted per range: msamd.h)
    if (r == 1)
      w->registered.push_back(const_cast<u32*>(traces[ci]));
    else if (r == 0)
      all_pinned = false;
  }
  w->h_claim_offs.assign(1, 0);
  for (size_t i = 0; i < n_claims; i++) {
    if (claim_offsets[i + 1] < claim_offsets[i]) throw std::runtime_error("claim offsets must not decrease");
    w->claims.emplace_back(claim_data + claim_offsets[i], claim_data + claim_offsets[i + 1]);
    for (u32 x : w->claims.back()) {
      if (x >= BB_P) throw std::runtime_error("non-canonical claim value");
      w->h_claims_monty.push_back(bb_to_monty(x));
    }
    w->h_claim_offs.push_back(claim_offsets[i + 1] - claim_offsets[0]);
  }
  if (pinned) *pinned = all_pinned;
  w->pinned = all_pinned;
  return w;
}

namespace {
// the per-proof upload of a host-resident witness: traces (converted to Montgomery form behind their copies) and claims
struct BHostUpload {
  BWitness& w;
  Ctx& ctx;
  BHostUpload(BWitness& wit, Ctx& c) : w(wit), ctx(c) {
    if (!w.host_resident) return;
    for (size_t ci = 0; ci < w.h_traces.size(); ci++)
      if (w.h_traces[ci]) {
        if (w.pinned)
          bb_upload_rows_async(ctx, w.h_traces[ci], w.heights[ci], w.sys->circuits[ci].main_width, w.traces[ci]);
        else
          bb_upload_rows(ctx, w.h_traces[ci], w.heights[ci], w.sys->circuits[ci].main_width, w.traces[ci]);  // (Ctx::bounce_h2d)
      }
    const size_t n_claims = w.claims.size();
    if (n_claims) {
      w.d_claim_offs = DBuf<u64>(ctx, n_claims + 1);
      w.d_claim_data = DBuf<u32>(ctx, std::max<size_t>(w.h_claims_monty.size(), 1));
      ctx.h2d(w.d_claim_offs.p, w.h_claim_offs.data(), (n_claims + 1) * 8);
      if (!w.h_claims_monty.empty()) ctx.h2d(w.d_claim_data.p, w.h_claims_monty.data(), w.h_claims_monty.size() * 4);
    }
  }
  ~BHostUpload() {
    if (!w.host_resident) return;
    (void)hipStreamSynchronize(ctx.stream);  // an abandoned proof may still be reading the blocks
    for (auto& t : w.traces) t = BMat();
    w.d_claim_offs.reset();
    w.d_claim_data.reset();
  }
};
}  // namespace

// ------------------------------------------------------------------ proof bytes (Proof::to_bytes, src/prover.rs:241-248)
namespace {
struct W {
  std::vector<uint8_t> b;
  void u8(uint8_t x) { b.push_back(x); }
  void u64_(u64 x) {
    for (int k = 0; k < 8; k++) b.push_back((uint8_t)(x >> (8 * k)));
  }
  void fe(u32 monty) {  // MontyField31 serialises its Montgomery word
    for (int k = 0; k < 4; k++) b.push_back((uint8_t)(monty >> (8 * k)));
  }
  void ext(E4 e) {
    for (int k = 0; k < 4; k++) fe(e.c[k]);
  }
  void dig(const Digest8& d) {
    for (int k = 0; k < 8; k++) fe(d.w[k]);
  }
  void cap(const std::vector<Digest8>& c) {
    u64_(c.size());
    for (auto& d : c) dig(d);
  }
};
typedef std::vector<std::vector<std::vector<E4>>> OpenedRound;  // matrix -> point -> column
void write_round(W& w, const OpenedRound& r) {
  w.u64_(r.size());
  for (auto& m : r) {
    w.u64_(m.size());
    for (auto& pt : m) {
      w.u64_(pt.size());
      for (auto& e : pt) w.ext(e);
    }
  }
}
struct OpenRound {
  const BPcsData* data;
  std::vector<std::vector<E4>> points;  // per matrix
};
struct E4Less {
  bool operator()(const E4& a, const E4& b) const {
    for (int k = 0; k < 4; k++)
      if (a.c[k] != b.c[k]) return a.c[k] < b.c[k];
    return false;
  }
};
struct FriOut {
  std::vector<std::vector<Digest8>> commits;
  std::vector<u32> pow_witnesses;  // canonical
  std::vector<E4> final_poly;
  u32 query_pow_witness = 0;
  std::vector<uint8_t> query_bytes;  // the serialised Vec<QueryProof> body (without its length)
};

// the opening_proof field of Proof::to_bytes (FriProof: commit-phase caps, their proof-of-work witnesses, the query proofs, the
// final polynomial, the query proof-of-work witness)
void write_fri(W& w, const FriOut& fri, const Params& prm) {
  w.u64_(fri.commits.size());
  for (auto& c : fri.commits) w.cap(c);
  w.u64_(fri.pow_witnesses.size());
  for (u32 x : fri.pow_witnesses) w.fe(bb_to_monty(x));
  w.u64_(prm.num_queries);
  w.b.insert(w.b.end(), fri.query_bytes.begin(), fri.query_bytes.end());
  w.u64_(fri.final_poly.size());
  for (auto& e : fri.final_poly) w.ext(e);
  w.fe(bb_to_monty(fri.query_pow_witness));
}

// challenger.grind(bits): from 8 bits on the search runs on the device (one permutation per candidate); the host
// challenger then replays the winning witness and checks it
u32 grind(BSystem& sys, Challenger& ch, unsigned bits) {
  if (bits < 8) return ch.grind(bits);
  if (ch.input.size() >= 8) throw std::runtime_error("grind: challenger queue full");
  u32 w = bb_grind(*sys.ctx, sys.d_perm.p, ch.state, ch.input.data(), (unsigned)ch.input.size(), bits);
  ch.observe(bb_to_monty(w));
  if (ch.sample_bits(bits) != 0) throw std::runtime_error("grind: device witness rejected by the host challenger");
  return w;
}

// TwoAdicFriPcs::open + prove_fri (p3-fri 0.5.1: open, prover::prove_fri / commit_phase / answer_query, TwoAdicFriFolding)
void pcs_open(BSystem& sys, const std::vector<OpenRound>& rounds, Challenger& ch, std::vector<OpenedRound>& opened, FriOut& fri) {
  Ctx& ctx = *sys.ctx;
  const Params& prm = sys.params;
  unsigned lb = (unsigned)prm.log_blowup;
  size_t gmax = 0, gw = 0;
  for (auto& r : rounds)
    for (auto& m : r.data->ldes) gmax = std::max(gmax, m.h), gw = std::max(gw, m.w);
  if (!gmax) throw std::runtime_error("no matrices supplied");
  unsigned log_gmax = log2_strict(gmax);

  // 1 / (z - x_i) and x_i / (z - x_i) per distinct point, for the tallest matrix opened there
  std::map<E4, size_t, E4Less> max_h;
  for (auto& r : rounds)
    for (size_t mi = 0; mi < r.data->ldes.size(); mi++)
      for (auto& z : r.points[mi]) {
        size_t& h = max_h[z];
        h = std::max(h, r.data->ldes[mi].h);
      }
  struct Denoms {
    DBuf<E4> inv, wgt;
  };
  std::map<E4, Denoms, E4Less> denoms;
  for (auto& kv : max_h) {
    Denoms d;
    d.inv = DBuf<E4>(ctx, kv.second);
    d.wgt = DBuf<E4>(ctx, kv.second);
    bb_inv_denoms(ctx, kv.first, log2_strict(kv.second), kv.second, d.inv.p, d.wgt.p);
    denoms.emplace(kv.first, std::move(d));
  }
  // opened values: barycentric interpolation over the first h = height / blowup storage rows (the coset g H_h); every
  // matrix and point is launched first, ONE read-back serves them all, then they are observed in round -> matrix -> point order
  u32 g = bb_to_monty(BB_GENERATOR);
  opened.clear();
  size_t n_part = 0;
  for (auto& r : rounds)
    for (size_t mi = 0; mi < r.data->ldes.size(); mi++) n_part += r.points[mi].size() * bb_bary_partials(r.data->ldes[mi].w, r.data->ldes[mi].h >> lb);
  DBuf<E4> d_part(ctx, std::max<size_t>(n_part, 1));
  {
    size_t off = 0;
    for (auto& r : rounds)
      for (size_t mi = 0; mi < r.data->ldes.size(); mi++) {
        const BMat& mat = r.data->ldes[mi];
        for (auto& z : r.points[mi]) {
          bb_bary_launch(ctx, mat, mat.h >> lb, denoms.at(z).wgt.p, d_part.p + off);
          off += bb_bary_partials(mat.w, mat.h >> lb);
        }
      }
  }
  std::vector<E4> h_part(std::max<size_t>(n_part, 1));
  ctx.d2h(h_part.data(), d_part.p, n_part * sizeof(E4));
  size_t part_off = 0;
  for (auto& r : rounds) {
    OpenedRound orr;
    for (size_t mi = 0; mi < r.data->ldes.size(); mi++) {
      const BMat& mat = r.data->ldes[mi];
      size_t h = mat.h >> lb;
      unsigned log_h = log2_strict(h);
      std::vector<std::vector<E4>> per_point;
      for (auto& z : r.points[mi]) {
        std::vector<E4> sums;
        bb_bary_finish(h_part.data() + part_off, mat.w, h, sums);
        part_off += bb_bary_partials(mat.w, h);
        u32 s_pow = bb_exp_pow2(g, log_h);
        E4 vanish = e4_exp_pow2(z, log_h);
        vanish.c[0] = bb_sub(vanish.c[0], s_pow);
        E4 scale = e4_mul_base(vanish, bb_inv(bb_mul(s_pow, bb_to_monty((u32)(h % BB_P)))));
        for (auto& y : sums) {
          y = e4_mul(y, scale);
          ch.observe_e4(y);
        }
        per_point.push_back(std::move(sums));
      }
      orr.push_back(std::move(per_point));
    }
    opened.push_back(std::move(orr));
  }

  E4 alpha = ch.sample_e4();
  std::vector<E4> apow(gw + 1);
  apow[0] = e4_one();
  for (size_t i = 1; i <= gw; i++) apow[i] = e4_mul(apow[i - 1], alpha);
  DBuf<E4> d_apow(ctx, apow.size());
  ctx.h2d(d_apow.p, apow.data(), apow.size() * sizeof(E4));

  // reduced openings per height (DEEP quotients)
  std::vector<size_t> num_reduced(33, 0);
  std::vector<DBuf<E4>> reduced(33);
  std::vector<size_t> reduced_len(33, 0);
  for (size_t ri = 0; ri < rounds.size(); ri++) {
    auto& r = rounds[ri];
    for (size_t mi = 0; mi < r.data->ldes.size(); mi++) {
      const BMat& mat = r.data->ldes[mi];
      unsigned lh = log2_strict(mat.h);
      if (!reduced_len[lh]) {
        reduced_len[lh] = mat.h;
        reduced[lh] = DBuf<E4>(ctx, mat.h);
        HIP_CHECK(hipMemsetAsync(reduced[lh].p, 0, mat.h * sizeof(E4), ctx.stream));
      }
      size_t np = r.points[mi].size();
      if (np == 0) continue;
      if (np > 2) throw std::runtime_error("more than two opening points per matrix");
      const E4* inv[2] = {nullptr, nullptr};
      E4 K[2], off[2];
      for (size_t pi = 0; pi < np; pi++) {
        const std::vector<E4>& ys = opened[ri][mi][pi];
        off[pi] = e4_pow(alpha, num_reduced[lh]);
        E4 red_z = e4_zero();
        for (size_t c = 0; c < mat.w; c++) red_z = e4_add(red_z, e4_mul(apow[c], ys[c]));
        K[pi] = e4_mul(off[pi], red_z);
        inv[pi] = denoms.at(r.points[mi][pi]).inv.p;
        num_reduced[lh] += mat.w;
      }
      bb_deep(ctx, mat, d_apow.p, (int)np, inv, K, off, reduced[lh].p);
    }
  }
  std::vector<std::pair<E4*, size_t>> inputs;
  for (int lh = 32; lh >= 0; lh--)
    if (reduced_len[lh]) inputs.emplace_back(reduced[lh].p, reduced_len[lh]);

  // ---- prove_fri: commit phase
  size_t final_len = size_t(1) << prm.log_final_poly_len;
  size_t stop = (size_t(1) << lb) * final_len;
  if (prm.log_final_poly_len > 0) {  // p3 prove_fri: every input must be taller than blowup * final length
    size_t min_h = inputs[0].second;
    for (auto& in : inputs) min_h = std::min(min_h, in.second);
    if (min_h <= stop) throw std::runtime_error("FRI: a committed matrix is not taller than blowup * final polynomial length");
  }
  std::vector<DBuf<E4>> layers;  // layers[i]: the vector folded in round i (its pairs are the leaves of FRI tree i)
  std::vector<BTree> fri_trees;
  std::vector<size_t> layer_len;
  E4* cur = inputs[0].first;
  size_t cur_len = inputs[0].second;
  size_t next_in = 1;
  unsigned log_max_height = log2_strict(cur_len);
  // Without proof of work the commit phase needs nothing from the host: with MSBB_DEV_FRI=1 the challenger steps run on
  // the device and the rounds queue up back to back; the host replays them afterwards. Off by default: measured at
  // 2^20 rows it does not pay (7.2 vs 6.9 ms) - the rounds are ~10 launches of ~5 us each, so the host's launch rate, not
  // its read-backs, is what spaces them; it needs the launches captured in a graph (or fewer of them) to win.
  // (round 2: the rounds are fused - one launch per round below 2^17 leaves - and the device transcript is the default;
  // MSBB_HOST_FRI=1 restores the host-driven rounds)
  // (rounds of arity above 2, max_log_arity > 1: on the device transcript too since the end of round 4, one round at a time - wide
  // leaves, tree and challenger step in bb_commit_pairs, then the round's folds with beta^(2^j) from the round's record; no
  // fused rounds. MSBB_HOST_WIDE_FRI=1: host-driven)
  const bool wide = prm.max_log_arity > 1;
  const bool dev_rounds = prm.commit_pow_bits == 0 && ch.input.size() < 8 && !(wide && getenv("MSBB_HOST_WIDE_FRI")) && !getenv("MSBB_HOST_FRI");
  const unsigned log_final_height = (unsigned)(prm.log_blowup + prm.log_final_poly_len);
  // p3-fri compute_log_arity_for_round: as far as max_log_arity allows without stepping over the next input or the final height
  auto round_arity = [&](size_t n, size_t ni) {
    const unsigned lh = log2_strict(n);
    unsigned la = std::min<unsigned>((unsigned)prm.max_log_arity, lh - log_final_height);
    if (ni < inputs.size()) la = std::min(la, lh - log2_strict(inputs[ni].second));
    if (la < 1) throw std::runtime_error("FRI: two inputs of one height");
    return la;
  };
  DBuf<DevChallenger> d_ch;
  DBuf<FriBeta> d_betas;
  size_t n_rounds = 0;  // (only used by the device transcript)
  {
    size_t ni = 1;
    for (size_t l = cur_len; l > stop;) {
      l >>= round_arity(l, ni);
      if (ni < inputs.size() && inputs[ni].second == l) ni++;
      n_rounds++;
    }
  }
  if (dev_rounds && n_rounds) {
    DevChallenger hc;
    for (int k = 0; k < 16; k++) hc.state[k] = ch.state[k];
    for (int k = 0; k < 8; k++) hc.input[k] = (size_t)k < ch.input.size() ? ch.input[k] : 0;
    hc.n_in = (u32)ch.input.size();
    hc.n_out = (u32)ch.output.size();  // host output buffer = state[..n_out] by construction (pops come off the back)
    d_ch = DBuf<DevChallenger>(ctx, 1);
    d_betas = DBuf<FriBeta>(ctx, n_rounds);
    ctx.h2d(d_ch.p, &hc, sizeof(hc));
  }
  size_t round = 0;
  bool tree_done = false;  // the previous round's fused launch already built this round's tree and ran its challenger step
  std::vector<unsigned> arities;  // log2 of every round's arity
  while (cur_len > stop) {
    const unsigned la = round_arity(cur_len, next_in);
    arities.push_back(la);
    size_t rows = cur_len >> la;
    if (!tree_done) {
      fri_trees.emplace_back();
      if (dev_rounds)  // the launch that produces the root also observes it and samples beta
        bb_commit_pairs(ctx, sys.d_perm.p, cur, rows, (unsigned)prm.cap_height, fri_trees.back(), d_ch.p, d_betas.p + round, la);
      else
        bb_commit_pairs(ctx, sys.d_perm.p, cur, rows, (unsigned)prm.cap_height, fri_trees.back(), nullptr, nullptr, la);
    }
    tree_done = false;
    const E4* roll = nullptr;
    if (next_in < inputs.size() && inputs[next_in].second == rows) roll = inputs[next_in++].first;
    DBuf<E4> out(ctx, rows);
    if (dev_rounds && wide) {
      const E4* src = cur;
      DBuf<E4> step;
      for (unsigned j = 0; j + 1 < la; j++) {
        DBuf<E4> nxt(ctx, cur_len >> (j + 1));
        bb_fri_fold_dev(ctx, src, cur_len >> (j + 1), d_betas.p + round, nullptr, nxt.p, j);
        step = std::move(nxt);
        src = step.p;
      }
      bb_fri_fold_dev(ctx, src, rows, d_betas.p + round, roll, out.p, la - 1);
    } else if (dev_rounds) {
      if (rows > stop && bb_fri_round_fusable(rows, (unsigned)prm.cap_height)) {
        // the next round - this fold, its leaf digests, its tree, its challenger step - is ONE launch
        fri_trees.emplace_back();
        bb_fri_round_fused(ctx, sys.d_perm.p, cur, rows, d_betas.p + round, roll, out.p, fri_trees.back(), d_ch.p, d_betas.p + round + 1);
        tree_done = true;
      } else {
        bb_fri_fold_dev(ctx, cur, rows, d_betas.p + round, roll, out.p);
      }
    } else {
      std::vector<Digest8> cap = tree_cap(ctx, fri_trees.back());
      ch.observe_cap(cap);
      fri.commits.push_back(cap);
      fri.pow_witnesses.push_back(grind(sys, ch, (unsigned)prm.commit_pow_bits));
      E4 beta = ch.sample_e4();
      // a round of arity 2^la is la binary folds with beta, beta^2, beta^4, ...; the vector rolled in behind it takes
      // beta^(2^la), the square of the last step's challenge (bb_fri_fold's own roll-in factor)
      const E4* src = cur;
      DBuf<E4> step;
      for (unsigned j = 0; j + 1 < la; j++) {
        DBuf<E4> nxt(ctx, cur_len >> (j + 1));
        bb_fri_fold(ctx, src, cur_len >> (j + 1), beta, nullptr, nxt.p);
        beta = e4_square(beta);
        step = std::move(nxt);
        src = step.p;
      }
      bb_fri_fold(ctx, src, rows, beta, roll, out.p);
    }
    // keep the folded-from vector alive for the query openings
    if (layers.empty()) {
      layers.emplace_back();  // layer 0 is inputs[0] (owned by `reduced`)
      layer_len.push_back(cur_len);
    }
    layers.push_back(std::move(out));
    layer_len.push_back(rows);
    cur = layers.back().p;
    cur_len = rows;
    round++;
  }
  if (dev_rounds && n_rounds) {  // one read-back: every cap and beta; the host challenger replays and checks
    std::vector<FriBeta> hb(n_rounds);
    std::vector<std::vector<Digest8>> caps(n_rounds);
    for (size_t i = 0; i < n_rounds; i++) {
      const BTree& t = fri_trees[i];
      caps[i].resize(t.sizes[t.cap_layer()]);
      ctx.d2h_queue(caps[i].data(), t.layers[t.cap_layer()].p, caps[i].size() * sizeof(Digest8));
    }
    ctx.d2h(hb.data(), d_betas.p, n_rounds * sizeof(FriBeta));
    for (size_t i = 0; i < n_rounds; i++) {
      ch.observe_cap(caps[i]);
      fri.commits.push_back(caps[i]);
      fri.pow_witnesses.push_back(0);
      if (!e4_eq(ch.sample_e4(), hb[i].beta)) throw std::runtime_error("FRI: device transcript diverged from the host challenger");
    }
  }
  if (next_in != inputs.size()) throw std::runtime_error("FRI: an input was never rolled in");
  // final polynomial: first final_len entries, undo the bit reversal, inverse DFT (tiny: on the host)
  {
    unsigned lf = (unsigned)prm.log_final_poly_len;
    std::vector<E4> head(final_len), nat(final_len);
    ctx.d2h(head.data(), cur, final_len * sizeof(E4));
    for (size_t i = 0; i < final_len; i++) nat[bitrev_host(i, lf)] = head[i];
    u32 w_inv = bb_inv(bb_two_adic_generator(lf)), n_inv = bb_inv(bb_to_monty((u32)final_len));
    fri.final_poly.resize(final_len);
    for (size_t k = 0; k < final_len; k++) {
      E4 acc = e4_zero();
      u32 step = bb_pow(w_inv, k), tw = BB_R1;
      for (size_t j = 0; j < final_len; j++) {
        acc = e4_add(acc, e4_mul_base(nat[j], tw));
        tw = bb_mul(tw, step);
      }
      fri.final_poly[k] = e4_mul_base(acc, n_inv);
      ch.observe_e4(fri.final_poly[k]);
    }
  }
  fri.query_pow_witness = grind(sys, ch, (unsigned)prm.query_pow_bits);

  // ---- query phase: one gather of every opened row, sibling value and authentication path
  std::vector<GatherSeg> segs;
  u32 pos = 0;
  auto add_seg = [&](const u32* src, u32 n, u32 stride) {
    segs.push_back(GatherSeg{src, pos, n, stride});
    u32 at = pos;
    pos += n;
    return at;
  };
  struct TreeOpen {
    std::vector<std::pair<u32, u32>> rows;  // (offset, width) per matrix
    u32 path_at, path_len;
  };
  auto open_tree_path = [&](const BTree& t, size_t index, TreeOpen& o) {
    unsigned log_max = log2_strict(t.sizes[0]);
    size_t ch_eff = std::min<size_t>(t.cap_height, t.layers.size() - 1);
    o.path_len = (u32)(log_max - ch_eff);
    o.path_at = pos;
    for (size_t i = 0; i + ch_eff < log_max; i++) add_seg((const u32*)(t.layers[i].p + ((index >> i) ^ 1)), 8, 1);
  };
  std::vector<size_t> indices;
  std::vector<std::vector<TreeOpen>> input_opens, fri_opens;
  std::vector<std::vector<u32>> sib_at;
  for (size_t qi = 0; qi < prm.num_queries; qi++) {
    size_t index = ch.sample_bits(log_max_height);
    indices.push_back(index);
    input_opens.emplace_back();
    for (auto& r : rounds) {
      const BTree& t = r.data->tree;
      unsigned lmh = log2_strict(t.sizes[0]);
      size_t idx = index >> (log_gmax - lmh);
      TreeOpen o;
      for (auto& m : r.data->ldes) {
        size_t row = idx >> (lmh - log2_strict(m.h));
        o.rows.emplace_back(add_seg(m.buf.p + row, (u32)m.w, (u32)m.ld), (u32)m.w);
      }
      open_tree_path(t, idx, o);
      input_opens.back().push_back(std::move(o));
    }
    fri_opens.emplace_back();
    sib_at.emplace_back();
    size_t index_i = index;
    for (size_t i = 0; i < fri_trees.size(); i++) {
      const unsigned la = arities[i];
      const size_t row_i = index_i >> la;
      const E4* vec = i == 0 ? inputs[0].first : layers[i].p;
      // binary round: the sibling value; wider: the whole row (the queried position's own value is dropped when the bytes are written)
      if (la == 1)
        sib_at.back().push_back(add_seg((const u32*)(vec + (index_i ^ 1)), 4, 1));
      else
        sib_at.back().push_back(add_seg((const u32*)(vec + (row_i << la)), 4u << la, 1));
      TreeOpen o;
      open_tree_path(fri_trees[i], row_i, o);
      fri_opens.back().push_back(std::move(o));
      index_i = row_i;
    }
  }
  std::vector<u32> g_out;
  bb_gather(ctx, segs, g_out);
  W w;
  for (size_t qi = 0; qi < prm.num_queries; qi++) {
    w.u64_(rounds.size());
    for (auto& o : input_opens[qi]) {
      w.u64_(o.rows.size());
      for (auto& rw : o.rows) {
        w.u64_(rw.second);
        for (u32 k = 0; k < rw.second; k++) w.fe(g_out[rw.first + k]);
      }
      w.u64_(o.path_len);
      for (u32 k = 0; k < o.path_len * 8; k++) w.fe(g_out[o.path_at + k]);
    }
    w.u64_(fri_trees.size());
    size_t index_i = indices[qi];
    for (size_t i = 0; i < fri_trees.size(); i++) {
      const unsigned la = arities[i];
      w.u8((uint8_t)la);  // log_arity
      w.u64_((size_t(1) << la) - 1);
      if (la == 1) {
        for (u32 k = 0; k < 4; k++) w.fe(g_out[sib_at[qi][i] + k]);
      } else {
        const size_t own = index_i & ((size_t(1) << la) - 1);
        for (size_t j = 0; j < (size_t(1) << la); j++)
          if (j != own)
            for (u32 k = 0; k < 4; k++) w.fe(g_out[sib_at[qi][i] + 4 * j + k]);
      }
      index_i >>= la;
      const TreeOpen& o = fri_opens[qi][i];
      w.u64_(o.path_len);
      for (u32 k = 0; k < o.path_len * 8; k++) w.fe(g_out[o.path_at + k]);
    }
  }
  fri.query_bytes = std::move(w.b);
}
}  // namespace

std::vector<uint8_t> prove(BSystem& sys, BWitness& wit, double* stage_ms) {
  Ctx& ctx = *sys.ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  double t_begin = now_ms(), t0;
  const Params& prm = sys.params;
  unsigned lb = (unsigned)prm.log_blowup;
  size_t C = sys.circuits.size();
  if (wit.traces.size() != C) throw std::runtime_error("witness/circuit count mismatch");
  BHostUpload upload(wit, ctx);  // host-resident witness: the uploads are queued now, in front of the transforms that read them
  Challenger ch(&sys.perm);
  for (u32 v : sys.seed) ch.observe(v);
  // System::observe_shape, src/system.rs:211-222
  ch.observe_usize(C);
  for (auto& c : sys.circuits) {
    ch.observe_usize(c.constraint_count), ch.observe_usize(c.max_constraint_degree), ch.observe_usize(c.pre_height);
    ch.observe_usize(c.pre_width), ch.observe_usize(c.main_width), ch.observe_usize(c.stage2_width);
  }
  std::vector<uint8_t> active;
  std::vector<size_t> active_idx;
  std::vector<int> active_pos(C, -1);
  for (size_t i = 0; i < C; i++) {
    bool a = wit.heights[i] > 0;
    active.push_back(a);
    ch.observe(a ? BB_R1 : 0);
    if (a) {
      active_pos[i] = (int)active_idx.size();
      active_idx.push_back(i);
    }
  }
  if (active_idx.empty()) throw std::runtime_error("cannot prove with every circuit deactivated");

  // ---- stage 1 commit
  t0 = now_ms();
  std::vector<unsigned> log_degrees;
  BPcsData s1;
  {
    std::vector<BMat> ldes;
    for (size_t ci : active_idx) {
      log_degrees.push_back(log2_strict(wit.heights[ci]));
      ldes.emplace_back();
      bb_coset_lde(ctx, wit.traces[ci], lb, ldes.back());
    }
    bb_commit(ctx, sys.d_perm.p, std::move(ldes), (unsigned)prm.cap_height, s1);
  }
  std::vector<Digest8> s1_cap = tree_cap(ctx, s1.tree);
  if (stage_ms) stage_ms[0] = now_ms() - t0;
  if (sys.has_pre) ch.observe_cap(sys.pre_commit);
  ch.observe_cap(s1_cap);
  for (unsigned ld : log_degrees) ch.observe_usize(ld);
  ch.observe_usize(wit.claims.size());
  for (auto& c : wit.claims) {
    ch.observe_usize(c.size());
    for (u32 x : c) ch.observe(bb_to_monty(x));
  }
  E4 beta = ch.sample_e4();
  ch.observe_e4(beta);
  E4 gamma = ch.sample_e4();
  ch.observe_e4(gamma);
  // claims accumulator, src/prover.rs:382-387
  E4 acc = bb_claims_accumulator(ctx, wit.d_claim_data.p, wit.d_claim_offs.p, wit.claims.size(), beta, gamma);

  // ---- lookup construction + stage 2 commit
  t0 = now_ms();
  std::vector<E4> accumulators;
  BPcsData s2;
  {
    std::vector<BMat> s2_traces;
    E4 running = acc;
    for (size_t ci : active_idx) {
      const BCircuit& c = sys.circuits[ci];
      s2_traces.emplace_back();
      E4 total;
      bb_stage2(ctx, c.prog, c.lookup_prefix_len, c.lk, wit.traces[ci], c.pre_width ? &c.pre : nullptr, beta, gamma, s2_traces.back(), &total);
      running = e4_add(running, total);
      accumulators.push_back(running);
    }
    if (stage_ms) stage_ms[1] = now_ms() - t0;
    t0 = now_ms();
    std::vector<BMat> ldes;
    for (auto& t : s2_traces) {
      ldes.emplace_back();
      bb_coset_lde(ctx, t, lb, ldes.back());
    }
    bb_commit(ctx, sys.d_perm.p, std::move(ldes), (unsigned)prm.cap_height, s2);
  }
  std::vector<Digest8> s2_cap = tree_cap(ctx, s2.tree);
  if (stage_ms) stage_ms[2] = now_ms() - t0;
  ch.observe_cap(s2_cap);
  for (auto& a : accumulators) ch.observe_e4(a);
  E4 alpha = ch.sample_e4();

  // ---- quotient
  t0 = now_ms();
  BPcsData qd;
  {
    std::vector<BMat> q_ldes;
    E4 cur_acc = acc;
    for (size_t pos = 0; pos < active_idx.size(); pos++) {
      size_t ci = active_idx[pos];
      const BCircuit& c = sys.circuits[ci];
      unsigned log_q = log2_strict(c.quotient_degree());
      BQuotientIn in;
      in.prog = &c.prog, in.lk = &c.lk, in.d_zeros = c.d_zeros.p, in.n_zeros = c.zeros.size(), in.constraint_count = c.constraint_count;
      in.jit = &c.quotient_jit;
      in.pre = sys.has_pre && sys.pre_indices[ci] >= 0 ? &sys.pre_data.ldes[sys.pre_indices[ci]] : nullptr;
      in.s1 = &s1.ldes[pos], in.s2 = &s2.ldes[pos];
      in.log_n = log_degrees[pos], in.log_q = log_q, in.log_blowup = lb;
      in.publics[0] = beta, in.publics[1] = gamma, in.publics[2] = cur_acc, in.publics[3] = accumulators[pos];
      in.alpha = alpha;
      BMat q_evals;
      bb_quotient(ctx, in, q_evals);
      q_ldes.emplace_back();
      bb_quotient_lde(ctx, q_evals, log_degrees[pos], log_q, lb, q_ldes.back());
      cur_acc = accumulators[pos];
    }
    bb_commit(ctx, sys.d_perm.p, std::move(q_ldes), (unsigned)prm.cap_height, qd);
  }
  std::vector<Digest8> q_cap = tree_cap(ctx, qd.tree);
  ch.observe_cap(q_cap);
  if (stage_ms) stage_ms[3] = now_ms() - t0;

  // ---- opening
  t0 = now_ms();
  E4 zeta = ch.sample_e4();
  std::vector<OpenRound> rounds(3);
  rounds[0].data = &s1, rounds[1].data = &s2, rounds[2].data = &qd;
  for (unsigned ld : log_degrees) {
    E4 zn = e4_mul_base(zeta, bb_two_adic_generator(ld));
    rounds[0].points.push_back({zeta, zn});
    rounds[1].points.push_back({zeta, zn});
    rounds[2].points.push_back({zeta});
  }
  if (sys.has_pre) {
    OpenRound r0;
    r0.data = &sys.pre_data;
    for (size_t ci = 0; ci < C; ci++) {
      if (sys.pre_indices[ci] < 0) continue;
      if (active_pos[ci] >= 0) {
        E4 zn = e4_mul_base(zeta, bb_two_adic_generator(log_degrees[active_pos[ci]]));
        r0.points.push_back({zeta, zn});
      } else {
        r0.points.push_back({});
      }
    }
    rounds.push_back(std::move(r0));
  }
  std::vector<OpenedRound> opened;
  FriOut fri;
  pcs_open(sys, rounds, ch, opened, fri);

  // ---- Proof::to_bytes
  W w;
  w.u64_(active.size());
  for (auto a : active) w.u8(a ? 1 : 0);
  w.cap(s1_cap), w.cap(s2_cap), w.cap(q_cap);
  w.u64_(accumulators.size());
  for (auto& e : accumulators) w.ext(e);
  w.u64_(log_degrees.size());
  for (auto d : log_degrees) w.u8((uint8_t)d);
  write_fri(w, fri, prm);
  write_round(w, opened[2]);
  w.u8(sys.has_pre ? 1 : 0);
  if (sys.has_pre) write_round(w, opened[3]);
  write_round(w, opened[0]);
  write_round(w, opened[1]);
  if (stage_ms) {
    stage_ms[4] = now_ms() - t0;
    stage_ms[5] = now_ms() - t_begin;
  }
  return std::move(w.b);
}


// ------------------------------------------------------------------ verify
// System::verify_multiple_claims (/root/reference/src/verifier.rs:208-532, shape checks :536-695) for this
// configuration, over the bytes msbb_prove writes. All host code (a verification is a few thousand permutations);
// written against the reference's verifier, independently of the oracle's. Codes: the VerificationError variants
// (src/verifier.rs:176-192) as in include/mstark.h.
namespace {
enum : int { V_OK = 0, V_INVALID_OPENING = 2, V_INVALID_SHAPE = 3, V_INVALID_SYSTEM = 4, V_OOD_MISMATCH = 5, V_UNBALANCED = 6 };
struct Malformed {};
struct PReader {
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
  size_t count(size_t elem_bytes) {
    u64 c = u64_();
    if (elem_bytes && c > (n - pos) / elem_bytes) throw Malformed();
    return (size_t)c;
  }
  u32 field() {  // the Montgomery word of a MontyField31; serde rejects words >= p
    need(4);
    u32 v;
    memcpy(&v, p + pos, 4);
    pos += 4;
    if (v >= BB_P) throw Malformed();
    return v;
  }
  E4 ext() {
    E4 e;
    for (int k = 0; k < 4; k++) e.c[k] = field();
    return e;
  }
  Digest8 digest() {
    Digest8 d;
    for (int k = 0; k < 8; k++) d.w[k] = field();
    return d;
  }
  std::vector<Digest8> cap() {
    std::vector<Digest8> v(count(32));
    for (auto& d : v) d = digest();
    return v;
  }
};
OpenedRound read_round(PReader& r) {
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
struct VBatchOpening {
  std::vector<std::vector<u32>> rows;
  std::vector<Digest8> path;
};
struct VFriStep {
  unsigned log_arity = 1;
  std::vector<E4> siblings;  // the opened row without the queried position's own value
  std::vector<Digest8> path;
};
struct VQuery {
  std::vector<VBatchOpening> inputs;
  std::vector<VFriStep> steps;
};
struct VProof {
  std::vector<uint8_t> active, log_degrees;
  std::vector<Digest8> s1, s2, q;
  std::vector<E4> accs;
  std::vector<std::vector<Digest8>> commits;
  std::vector<u32> pow;
  std::vector<VQuery> queries;
  std::vector<E4> final_poly;
  u32 query_pow = 0;
  OpenedRound q_opened, pre_opened, s1_opened, s2_opened;
  bool has_pre = false;
};
VProof parse_proof(const uint8_t* bytes, size_t len) {
  PReader r{bytes, len};
  VProof p;
  p.active.resize(r.count(1));
  for (auto& a : p.active) {
    a = r.u8();
    if (a > 1) throw Malformed();
  }
  p.s1 = r.cap(), p.s2 = r.cap(), p.q = r.cap();
  p.accs.resize(r.count(16));
  for (auto& a : p.accs) a = r.ext();
  p.log_degrees.resize(r.count(1));
  for (auto& l : p.log_degrees) l = r.u8();
  p.commits.resize(r.count(8));
  for (auto& c : p.commits) c = r.cap();
  p.pow.resize(r.count(4));
  for (auto& w : p.pow) w = r.field();
  p.queries.resize(r.count(8));
  for (auto& q : p.queries) {
    q.inputs.resize(r.count(8));
    for (auto& bo : q.inputs) {
      bo.rows.resize(r.count(8));
      for (auto& row : bo.rows) {
        row.resize(r.count(4));
        for (auto& v : row) v = r.field();
      }
      bo.path.resize(r.count(32));
      for (auto& d : bo.path) d = r.digest();
    }
    q.steps.resize(r.count(8));
    for (auto& st : q.steps) {
      st.log_arity = r.u8();
      if (st.log_arity < 1 || st.log_arity > BB_FRI_MAX_LOG_ARITY) throw Malformed();
      st.siblings.resize(r.count(16));
      if (st.siblings.size() != (size_t(1) << st.log_arity) - 1) throw Malformed();
      for (auto& e : st.siblings) e = r.ext();
      st.path.resize(r.count(32));
      for (auto& d : st.path) d = r.digest();
    }
  }
  p.final_poly.resize(r.count(16));
  for (auto& e : p.final_poly) e = r.ext();
  p.query_pow = r.field();
  p.q_opened = read_round(r);
  {
    const uint8_t tag = r.u8();
    if (tag > 1) throw Malformed();
    p.has_pre = tag != 0;
  }
  if (p.has_pre) p.pre_opened = read_round(r);
  p.s1_opened = read_round(r);
  p.s2_opened = read_round(r);
  if (r.pos != len) throw Malformed();
  return p;
}
bool e4_is_zero(E4 a) { return !(a.c[0] | a.c[1] | a.c[2] | a.c[3]); }

// PaddingFreeSponge / TruncatedPermutation on the host
Digest8 hash_words(const Poseidon2& perm, const std::vector<u32>& v) {
  u32 st[16] = {0};
  for (size_t i = 0; i < v.size(); i += 8) {
    size_t k = std::min<size_t>(8, v.size() - i);
    for (size_t j = 0; j < k; j++) st[j] = v[i + j];
    bb_poseidon2(perm, st);
  }
  Digest8 d;
  for (int j = 0; j < 8; j++) d.w[j] = st[j];
  return d;
}
Digest8 compress_host(const Poseidon2& perm, const Digest8& l, const Digest8& r) {
  u32 st[16];
  for (int j = 0; j < 8; j++) st[j] = l.w[j], st[8 + j] = r.w[j];
  bb_poseidon2(perm, st);
  Digest8 d;
  for (int j = 0; j < 8; j++) d.w[j] = st[j];
  return d;
}
struct Dim {
  size_t w, h;
};
bool mmcs_verify_batch(const Poseidon2& perm, const std::vector<Digest8>& cap, const std::vector<Dim>& dims, size_t index, const VBatchOpening& o) {
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
    std::vector<u32> buf;
    while (pos < order.size() && dims[order[pos]].h == height) {
      auto& v = o.rows[order[pos]];
      buf.insert(buf.end(), v.begin(), v.end());
      pos++;
    }
    return hash_words(perm, buf);
  };
  Digest8 root = hash_group(cur);
  const size_t capn = cap.size();
  if (capn == 0 || (capn & (capn - 1))) return false;
  const unsigned chh = log2_strict(capn);
  if (chh > log_max || o.path.size() != log_max - chh) return false;
  size_t idx = index;
  if (idx >= (size_t(1) << log_max)) return false;
  for (auto& sib : o.path) {
    root = (idx & 1) ? compress_host(perm, sib, root) : compress_host(perm, root, sib);
    idx >>= 1;
    cur >>= 1;
    if (pos < order.size() && dims[order[pos]].h == cur) root = compress_host(perm, root, hash_group(cur));
  }
  if (pos != order.size()) return false;
  return memcmp(root.w, cap[idx].w, 32) == 0;
}
bool check_witness(Challenger& ch, unsigned bits, u32 monty_witness) {
  if (bits == 0) return true;
  ch.observe(monty_witness);
  return ch.sample_bits(bits) == 0;
}
struct RoundClaim {
  std::vector<Digest8> commit;
  std::vector<unsigned> log_n;
  std::vector<std::vector<std::pair<E4, const std::vector<E4>*>>> mats;
};
// TwoAdicFriPcs::verify + verify_fri / verify_query (p3-fri 0.5.1)
bool pcs_verify(const BSystem& sys, const std::vector<RoundClaim>& rounds, const VProof& proof, Challenger& ch) {
  const Params& prm = sys.params;
  const Poseidon2& perm = sys.perm;
  const unsigned lb = (unsigned)prm.log_blowup;
  for (auto& r : rounds)
    for (auto& m : r.mats)
      for (auto& pv : m)
        for (auto& y : *pv.second) ch.observe_e4(y);
  const E4 alpha = ch.sample_e4();
  const size_t nrounds = proof.commits.size();
  if (proof.pow.size() != nrounds) return false;
  // every query repeats the rounds' arities; the first one's place the tallest input, each is checked against the schedule below
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
  if (log_gmax > BB_TWO_ADICITY) return false;
  std::vector<E4> betas;
  for (size_t i = 0; i < nrounds; i++) {
    ch.observe_cap(proof.commits[i]);
    if (!check_witness(ch, (unsigned)prm.commit_pow_bits, proof.pow[i])) return false;
    betas.push_back(ch.sample_e4());
  }
  if (proof.final_poly.size() != (size_t(1) << prm.log_final_poly_len)) return false;
  for (auto& c : proof.final_poly) ch.observe_e4(c);
  if (proof.queries.size() != prm.num_queries) return false;
  if (!check_witness(ch, (unsigned)prm.query_pow_bits, proof.query_pow)) return false;
  const unsigned log_final_height = (unsigned)(lb + prm.log_final_poly_len);
  const u32 g = bb_to_monty(BB_GENERATOR);
  for (auto& qp : proof.queries) {
    const size_t index = ch.sample_bits(log_gmax);
    if (qp.inputs.size() != rounds.size()) return false;
    std::map<unsigned, std::pair<E4, E4>> ro;  // log height -> (running alpha power, reduced opening)
    for (size_t ri = 0; ri < rounds.size(); ri++) {
      const RoundClaim& r = rounds[ri];
      const VBatchOpening& bo = qp.inputs[ri];
      if (bo.rows.size() != r.mats.size()) return false;
      std::vector<Dim> dims;
      unsigned log_bmax = 0;
      for (size_t mi = 0; mi < r.mats.size(); mi++) {
        dims.push_back(Dim{bo.rows[mi].size(), size_t(1) << (r.log_n[mi] + lb)});
        log_bmax = std::max(log_bmax, r.log_n[mi] + lb);
      }
      if (log_bmax > log_gmax) return false;
      if (!mmcs_verify_batch(perm, r.commit, dims, index >> (log_gmax - log_bmax), bo)) return false;
      for (size_t mi = 0; mi < r.mats.size(); mi++) {
        const unsigned lh = r.log_n[mi] + lb;
        const size_t rev = bitrev_host(index >> (log_gmax - lh), lh);
        const u32 x = bb_mul(g, bb_pow(bb_two_adic_generator(lh), rev));
        auto it = ro.find(lh);
        if (it == ro.end()) it = ro.emplace(lh, std::make_pair(e4_one(), e4_zero())).first;
        for (auto& pv : r.mats[mi]) {
          if (pv.second->size() != bo.rows[mi].size()) return false;
          E4 den = pv.first;
          den.c[0] = bb_sub(den.c[0], x);
          if (e4_is_zero(den)) return false;
          const E4 quot = e4_inv(den);
          for (size_t c = 0; c < pv.second->size(); c++) {
            E4 diff = (*pv.second)[c];
            diff.c[0] = bb_sub(diff.c[0], bo.rows[mi][c]);
            it->second.second = e4_add(it->second.second, e4_mul(e4_mul(it->second.first, diff), quot));
            it->second.first = e4_mul(it->second.first, alpha);
          }
        }
      }
    }
    auto low = ro.find(lb);  // a height-1 trace gives a constant polynomial: its reduced opening must vanish
    if (low != ro.end() && log_final_height >= lb && lb < log_gmax) {
      if (!e4_is_zero(low->second.second)) return false;
      ro.erase(low);
    }
    if (qp.steps.size() != nrounds) return false;
    auto it = ro.rbegin();
    if (it == ro.rend() || it->first != log_gmax) return false;
    E4 folded = it->second.second;
    ++it;
    size_t idx = index;
    unsigned log_height = log_gmax;
    for (size_t i = 0; i < nrounds; i++) {
      const VFriStep& st = qp.steps[i];
      const unsigned la = st.log_arity;
      if (la != arities[i] || log_height <= log_final_height) return false;
      {  // the schedule: as far as max_log_arity allows without stepping over the next input or below the final height
        unsigned want = std::min<unsigned>((unsigned)prm.max_log_arity, log_height - log_final_height);
        if (it != ro.rend()) want = std::min(want, log_height - it->first);
        if (la != want) return false;
      }
      const unsigned log_folded_height = log_height - la;
      const size_t m = size_t(1) << la, own = idx & (m - 1), row = idx >> la;
      std::vector<E4> evals(m);
      for (size_t j = 0, k = 0; j < m; j++) evals[j] = j == own ? folded : st.siblings[k++];
      VBatchOpening bo;
      bo.rows.emplace_back();
      for (auto& e : evals)
        for (int k = 0; k < 4; k++) bo.rows[0].push_back(e.c[k]);  // ExtensionMmcs: flattened row
      bo.path = st.path;
      if (!mmcs_verify_batch(perm, proof.commits[i], {Dim{4 * m, size_t(1) << log_folded_height}}, row, bo)) return false;
      idx = row;
      if (la == 1) {
        // fold_row: the line through (x0, e0), (-x0, e1) evaluated at beta; x0 = w^bitrev(idx) on the subgroup
        const u32 x0 = bb_pow(bb_two_adic_generator(log_folded_height + 1), bitrev_host(idx, log_folded_height));
        const u32 x1 = bb_neg(x0);
        const E4 slope = e4_mul_base(e4_sub(evals[1], evals[0]), bb_inv(bb_sub(x1, x0)));
        E4 bx = betas[i];
        bx.c[0] = bb_sub(bx.c[0], x0);
        folded = e4_add(evals[0], e4_mul(bx, slope));
      } else {
        // position j of the row holds the value at h_j = x w^bitrev(j), w of order m = 2^la, x = w_{2^log_height}^bitrev(row); fold_row
        // is the polynomial of degree < m through them at beta, in barycentric form over the coset x <w>:
        // p(beta) = (beta^m - x^m) / (m x^m) * sum_j e_j h_j / (beta - h_j)
        const u32 x = bb_pow(bb_two_adic_generator(log_height), bitrev_host(row, log_folded_height));
        const u32 wm = bb_two_adic_generator(la);
        E4 sum = e4_zero();
        bool hit = false;
        for (size_t j = 0; j < m && !hit; j++) {
          const u32 h = bb_mul(x, bb_pow(wm, bitrev_host(j, la)));
          E4 d = betas[i];
          d.c[0] = bb_sub(d.c[0], h);
          if (e4_is_zero(d)) {  // beta is one of the row's points
            folded = evals[j];
            hit = true;
          } else {
            sum = e4_add(sum, e4_mul(e4_mul_base(evals[j], h), e4_inv(d)));
          }
        }
        if (!hit) {
          const u32 xm = bb_pow(x, m);
          E4 z = betas[i];
          for (unsigned k = 0; k < la; k++) z = e4_square(z);
          z.c[0] = bb_sub(z.c[0], xm);
          folded = e4_mul(e4_mul_base(z, bb_inv(bb_mul(xm, bb_to_monty((u32)m)))), sum);
        }
      }
      log_height = log_folded_height;
      if (it != ro.rend() && it->first == log_folded_height) {
        // roll-in factor: the next power of beta after the 2^la the fold used (beta^2 for a binary round)
        E4 f = betas[i];
        for (unsigned k = 0; k < la; k++) f = e4_square(f);
        folded = e4_add(folded, e4_mul(f, it->second.second));
        ++it;
      }
    }
    if (it != ro.rend()) return false;
    const u32 x = bb_pow(bb_two_adic_generator(log_gmax), bitrev_host(idx, log_gmax));
    E4 eval = e4_zero();
    for (size_t k = proof.final_poly.size(); k-- > 0;) eval = e4_add(e4_mul_base(eval, x), proof.final_poly[k]);
    if (!e4_eq(eval, folded)) return false;
  }
  return true;
}
// src/lookup.rs:103-118 over extension-valued coordinates
void coord_mul_e(const E4* a, const E4* b, E4* out) {
  E4 lo[4], hi[4];
  for (int k = 0; k < 4; k++) lo[k] = hi[k] = e4_zero();
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      E4 p = e4_mul(a[i], b[j]);
      if (i + j < 4)
        lo[i + j] = e4_add(lo[i + j], p);
      else
        hi[i + j - 4] = e4_add(hi[i + j - 4], p);
    }
  const u32 w = bb_to_monty(BB_EXT_W);
  for (int k = 0; k < 4; k++) out[k] = e4_add(lo[k], e4_mul_base(hi[k], w));
}
}  // namespace

int verify(BSystem& sys, size_t n_claims, const u64* claim_offsets, const u32* claim_data, const uint8_t* proof_bytes, size_t proof_len) {
  const Params& prm = sys.params;
  const size_t C = sys.circuits.size();
  if (C == 0) return V_INVALID_SYSTEM;
  VProof proof;
  try {
    proof = parse_proof(proof_bytes, proof_len);
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
    const BCircuit& c = sys.circuits[ci];
    const int slot = sys.pre_indices[ci];
    if (proof.s1_opened[pos].size() != 2 || proof.s2_opened[pos].size() != 2) return V_INVALID_SHAPE;
    if (slot >= 0 && proof.pre_opened[slot].size() != 2) return V_INVALID_SHAPE;
    for (int j = 0; j < 2; j++) {
      if (slot >= 0 && proof.pre_opened[slot][j].size() != c.pre_width) return V_INVALID_SHAPE;
      if (proof.s1_opened[pos][j].size() != c.main_width) return V_INVALID_SHAPE;
      if (proof.s2_opened[pos][j].size() != c.stage2_width) return V_INVALID_SHAPE;
    }
    const size_t qd = c.quotient_degree();
    if (proof.log_degrees[pos] + log2_strict(qd) > BB_TWO_ADICITY - prm.log_blowup) return V_INVALID_SHAPE;  // baby_bear_config.rs:87
    if (c.pre_width && (size_t(1) << proof.log_degrees[pos]) != c.pre_height) return V_INVALID_SHAPE;
    qdeg.push_back(qd);
    if (proof.q_opened[pos].size() != 1 || proof.q_opened[pos][0].size() != qd * 4) return V_INVALID_SHAPE;
  }
  if (proof.accs.size() != na) return V_INVALID_SHAPE;
  if (!e4_is_zero(proof.accs.back())) return V_UNBALANCED;  // src/verifier.rs:242-246

  // ---- transcript replay (src/verifier.rs:255-326)
  for (size_t i = 0; i < n_claims; i++)
    if (claim_offsets[i + 1] < claim_offsets[i]) return V_INVALID_SHAPE;
  const size_t claim_elems = n_claims ? (size_t)claim_offsets[n_claims] : 0;
  for (size_t i = 0; i < claim_elems; i++)
    if (claim_data[i] >= BB_P) return V_INVALID_SHAPE;
  Challenger ch(&sys.perm);
  for (u32 v : sys.seed) ch.observe(v);
  ch.observe_usize(C);
  for (auto& c : sys.circuits) {
    ch.observe_usize(c.constraint_count), ch.observe_usize(c.max_constraint_degree), ch.observe_usize(c.pre_height);
    ch.observe_usize(c.pre_width), ch.observe_usize(c.main_width), ch.observe_usize(c.stage2_width);
  }
  for (auto a : proof.active) ch.observe(a ? BB_R1 : 0);
  if (sys.has_pre) ch.observe_cap(sys.pre_commit);
  ch.observe_cap(proof.s1);
  for (auto ld : proof.log_degrees) ch.observe_usize(ld);
  ch.observe_usize(n_claims);
  for (size_t i = 0; i < n_claims; i++) {
    ch.observe_usize(claim_offsets[i + 1] - claim_offsets[i]);
    for (u64 k = claim_offsets[i]; k < claim_offsets[i + 1]; k++) ch.observe(bb_to_monty(claim_data[k]));
  }
  const E4 beta = ch.sample_e4();
  ch.observe_e4(beta);
  const E4 gamma = ch.sample_e4();
  ch.observe_e4(gamma);
  ch.observe_cap(proof.s2);
  for (auto& a : proof.accs) ch.observe_e4(a);
  E4 acc = e4_zero();
  for (size_t i = 0; i < n_claims; i++) {
    E4 f = e4_zero();
    for (u64 k = claim_offsets[i + 1]; k-- > claim_offsets[i];) {
      f = e4_mul(f, gamma);
      f.c[0] = bb_add(f.c[0], bb_to_monty(claim_data[k]));
    }
    const E4 m = e4_add(beta, f);
    if (e4_is_zero(m)) return V_INVALID_SHAPE;  // the reference would divide by zero here
    acc = e4_add(acc, e4_inv(m));
  }
  const E4 alpha = ch.sample_e4();
  ch.observe_cap(proof.q);
  const E4 zeta = ch.sample_e4();

  std::vector<RoundClaim> rounds(3);
  rounds[0].commit = proof.s1, rounds[1].commit = proof.s2, rounds[2].commit = proof.q;
  for (size_t pos = 0; pos < na; pos++) {
    const unsigned ld = proof.log_degrees[pos];
    const E4 zn = e4_mul_base(zeta, bb_two_adic_generator(ld));
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
        const E4 zn = e4_mul_base(zeta, bb_two_adic_generator(ld));
        r0.log_n.push_back(ld);
        r0.mats.push_back({{zeta, &proof.pre_opened[slot][0]}, {zn, &proof.pre_opened[slot][1]}});
      } else {
        r0.log_n.push_back(log2_strict(sys.circuits[ci].pre_height));
        r0.mats.push_back({});
      }
    }
    rounds.push_back(std::move(r0));
  }
  if (!pcs_verify(sys, rounds, proof, ch)) return V_INVALID_OPENING;

  // ---- out-of-domain check per circuit (src/verifier.rs:419-530)
  for (size_t pos = 0; pos < na; pos++) {
    const size_t ci = aidx[pos];
    const BCircuit& c = sys.circuits[ci];
    const unsigned ld = proof.log_degrees[pos];
    const E4 next_acc = proof.accs[pos];
    const u32 g_n = bb_two_adic_generator(ld), g_inv = bb_inv(g_n);
    E4 zh = e4_exp_pow2(zeta, ld);  // selectors_at_point
    zh.c[0] = bb_sub(zh.c[0], BB_R1);
    E4 z1 = zeta, zg = zeta;
    z1.c[0] = bb_sub(z1.c[0], BB_R1);
    zg.c[0] = bb_sub(zg.c[0], g_inv);
    if (e4_is_zero(zh) || e4_is_zero(z1) || e4_is_zero(zg)) return V_OOD_MISMATCH;
    const E4 is_first = e4_mul(zh, e4_inv(z1)), is_last = e4_mul(zh, e4_inv(zg)), is_trans = zg, inv_van = e4_inv(zh);
    const u32 inj_norm = bb_inv(bb_mul(bb_to_monty((u32)((u64(1) << ld) % BB_P)), g_n));
    const E4 four[4] = {beta, gamma, acc, next_acc};
    E4 publics[16];
    for (int k = 0; k < 4; k++)
      for (int d = 0; d < 4; d++) publics[4 * k + d] = e4_base(four[k].c[d]);
    const int slot = sys.pre_indices[ci];
    const std::vector<E4>* rows[3][2] = {{slot >= 0 ? &proof.pre_opened[slot][0] : nullptr, slot >= 0 ? &proof.pre_opened[slot][1] : nullptr},
                                         {&proof.s1_opened[pos][0], &proof.s1_opened[pos][1]},
                                         {&proof.s2_opened[pos][0], &proof.s2_opened[pos][1]}};
    std::vector<E4> buf(c.nodes.size());
    for (size_t i = 0; i < c.nodes.size(); i++) {  // ConstraintGraph::sweep_range over the extension field
      const PNode& n = c.nodes[i];
      E4 v;
      switch (n.kind) {
        case msamd::OP_CONST: v = e4_base(bb_to_monty((u32)n.a)); break;
        case msamd::OP_VAR: {
          if (n.source > 2 || n.offset > 1) return V_INVALID_SYSTEM;
          const std::vector<E4>* row = rows[n.source][n.offset];
          if (!row || n.a >= row->size()) return V_INVALID_SYSTEM;
          v = (*row)[n.a];
          break;
        }
        case msamd::OP_PUBLIC:
          if (n.a >= 16) return V_INVALID_SYSTEM;
          v = publics[n.a];
          break;
        case msamd::OP_IS_FIRST: v = is_first; break;
        case msamd::OP_IS_LAST: v = is_last; break;
        case msamd::OP_IS_TRANS: v = is_trans; break;
        case msamd::OP_ADD: v = e4_add(buf[n.a], buf[n.b]); break;
        case msamd::OP_SUB: v = e4_sub(buf[n.a], buf[n.b]); break;
        case msamd::OP_MUL: v = e4_mul(buf[n.a], buf[n.b]); break;
        default: v = e4_neg(buf[n.a]); break;
      }
      buf[i] = v;
    }
    std::vector<E4> cv;
    for (auto z : c.zeros) cv.push_back(buf[z]);
    // logup_constraint_values, generic-degree path (src/lookup.rs:210-256)
    const std::vector<E4>&s2 = proof.s2_opened[pos][0], &s2n = proof.s2_opened[pos][1];
    E4 inj[4];
    for (int d = 0; d < 4; d++) inj[d] = e4_mul(is_last, e4_mul_base(e4_sub(publics[12 + d], publics[8 + d]), inj_norm));
    if (c.lookups.empty()) {
      for (int d = 0; d < 4; d++) cv.push_back(e4_add(e4_sub(s2n[d], s2[d]), inj[d]));
    } else {
      const size_t last = c.lookups.size() - 1;
      for (size_t j = 0; j < c.lookups.size(); j++) {
        const auto& l = c.lookups[j];
        E4 diff[4], f[4], t[4];
        for (int d = 0; d < 4; d++) {
          const E4 tgt = j < last ? s2[4 * (j + 1) + d] : e4_add(s2n[d], inj[d]);
          diff[d] = e4_sub(tgt, s2[4 * j + d]);
          f[d] = e4_zero();
        }
        for (size_t k = l.second.size(); k-- > 0;) {
          coord_mul_e(f, publics + 4, t);
          for (int d = 0; d < 4; d++) f[d] = t[d];
          f[0] = e4_add(f[0], buf[l.second[k]]);
        }
        for (int d = 0; d < 4; d++) f[d] = e4_add(f[d], publics[d]);
        coord_mul_e(f, diff, t);
        cv.push_back(e4_sub(t[0], buf[l.first]));
        for (int d = 1; d < 4; d++) cv.push_back(t[d]);
      }
    }
    if (cv.size() != c.constraint_count) return V_INVALID_SYSTEM;
    E4 comp = e4_zero();
    for (auto& x : cv) comp = e4_add(e4_mul(comp, alpha), x);
    // Q(zeta) = sum_i zeta^(i n) c_i(zeta), each chunk given by its four base-field coordinate polynomials
    const std::vector<E4>& qrow = proof.q_opened[pos][0];
    const E4 zpn = e4_exp_pow2(zeta, ld);
    E4 zp = e4_one(), quot = e4_zero();
    for (size_t i = 0; i < qdeg[pos]; i++) {
      E4 chunk = e4_zero();
      for (int d = 0; d < 4; d++) {
        E4 basis = e4_zero();
        basis.c[d] = BB_R1;  // X^d
        chunk = e4_add(chunk, e4_mul(qrow[4 * i + d], basis));
      }
      quot = e4_add(quot, e4_mul(zp, chunk));
      zp = e4_mul(zp, zpn);
    }
    if (!e4_eq(e4_mul(comp, inv_van), quot)) return V_OOD_MISMATCH;
    acc = next_acc;
  }
  return V_OK;
}

}  // namespace msbb

// ------------------------------------------------------------------ C ABI (include/mstark_bb.h)
using namespace msbb;

namespace msamd {  // capi.hip
void set_last_error(const char* what);
Ctx* ctx_of(ms_ctx* c);
void ctx_retain(ms_ctx* c);
void ctx_release(ms_ctx* c);
}  // namespace msamd
// Handle lifetimes as in include/mstark.h: a system keeps its context alive, a witness its system, an mmcs its context.
struct msbb_system {
  ms_ctx* owner = nullptr;
  std::unique_ptr<BSystem> sys;
  int refs = 1;
};
static void system_unref(msbb_system* s) {
  if (s && --s->refs == 0) {
    ms_ctx* c = s->owner;
    s->sys.reset();
    delete s;
    msamd::ctx_release(c);
  }
}
struct msbb_witness {
  msbb_system* owner = nullptr;
  std::unique_ptr<BWitness> w;
};
struct msbb_mmcs {
  ms_ctx* owner = nullptr;
  Ctx* ctx = nullptr;
  BPcsData data;
  // Level 2: a view of prover data owned by a system (its preprocessed commitment), which the handle keeps alive
  const BPcsData* view = nullptr;
  msbb_system* sys_ref = nullptr;
  const BPcsData& d() const { return view ? *view : data; }
};
// Level 2: a matrix that stays in HBM between two steps (stage-2 evaluations, a quotient LDE)
struct msbb_trace {
  ms_ctx* owner = nullptr;
  BMat m;
  int kind = 0;  // 0 evaluations over the trace domain (natural order), 1 committed-domain LDE (bit-reversed storage)
};
struct msbb_challenger {
  msbb_system* owner = nullptr;
  std::unique_ptr<Challenger> ch;
};
// a process-wide permutation for the PCS-level entry points (msbb_set_poseidon2)
struct PermHolder {
  Poseidon2 host;
  Poseidon2* dev = nullptr;
  bool set = false;
};
// one permutation per device (msbb_set_poseidon2 on a context of that device); contexts may live on different threads
static std::mutex g_perm_mu;
static PermHolder& perm_holder(int device) {
  static std::map<int, PermHolder> h;
  std::lock_guard<std::mutex> lock(g_perm_mu);
  return h[device];
}

#define BB_TRY try {
#define BB_CATCH                                  \
  }                                               \
  catch (const std::exception& e) {               \
    msamd::set_last_error(e.what());              \
    return MS_ERR;                                \
  }                                               \
  catch (...) {                                   \
    msamd::set_last_error("unknown error");       \
    return MS_ERR;                                \
  }

extern "C" {

int32_t msbb_system_create(ms_ctx* ctx, const uint8_t* blob, size_t len, msbb_system** out) {
  BB_TRY
  if (!ctx || !blob || !out) throw std::runtime_error("null argument");
  std::unique_ptr<msbb_system> s(new msbb_system());
  s->sys = system_from_blob(*msamd::ctx_of(ctx), blob, len);
  s->owner = ctx;
  msamd::ctx_retain(ctx);
  *out = s.release();
  return MS_OK;
  BB_CATCH
}
void msbb_system_destroy(msbb_system* sys) { system_unref(sys); }
int32_t msbb_system_preprocessed_commit(const msbb_system* sys, uint32_t* out, size_t cap_words, size_t* n_digests) {
  BB_TRY
  if (!sys || !n_digests) throw std::runtime_error("null argument");
  const BSystem& s = *sys->sys;
  *n_digests = s.has_pre ? s.pre_commit.size() : 0;
  if (*n_digests * 8 > cap_words) return MS_ERR_BUFFER;
  for (size_t i = 0; i < *n_digests; i++)
    for (int k = 0; k < 8; k++) out[8 * i + k] = bb_from_monty(s.pre_commit[i].w[k]);
  return MS_OK;
  BB_CATCH
}
int32_t msbb_system_circuit_info(const msbb_system* sys, size_t circuit, uint64_t out9[9]) {
  BB_TRY
  if (!sys || circuit >= sys->sys->circuits.size()) throw std::runtime_error("circuit index out of range");
  const BCircuit& c = sys->sys->circuits[circuit];
  const uint64_t v[9] = {c.main_width,       c.pre_width,           c.pre_height,         c.num_lookups, c.stage2_width,
                         c.constraint_count, c.max_constraint_degree, c.quotient_degree(), c.args_width};
  for (int i = 0; i < 9; i++) out9[i] = v[i];
  return MS_OK;
  BB_CATCH
}
int32_t msbb_witness_create(msbb_system* sys, const uint32_t* const* traces, const uint64_t* heights, size_t n_claims,
                            const uint64_t* claim_offsets, const uint32_t* claim_data, msbb_witness** out) {
  BB_TRY
  if (!sys || !traces || !heights || !out) throw std::runtime_error("null argument");
  std::unique_ptr<msbb_witness> w(new msbb_witness());
  w->w = witness_create(*sys->sys, traces, heights, n_claims, claim_offsets, claim_data);
  w->owner = sys;
  sys->refs++;
  *out = w.release();
  return MS_OK;
  BB_CATCH
}
int32_t msbb_witness_create_host(msbb_system* sys, const uint32_t* const* traces, const uint64_t* heights, size_t n_claims,
                                 const uint64_t* claim_offsets, const uint32_t* claim_data, int32_t* pinned, msbb_witness** out) {
  BB_TRY
  if (!sys || !traces || !heights || !out) throw std::runtime_error("null argument");
  std::unique_ptr<msbb_witness> w(new msbb_witness());
  bool all = false;
  w->w = witness_create_host(*sys->sys, traces, heights, n_claims, claim_offsets, claim_data, &all);
  if (pinned) *pinned = all ? 1 : 0;
  w->owner = sys;
  sys->refs++;
  *out = w.release();
  return MS_OK;
  BB_CATCH
}
void msbb_witness_destroy(msbb_witness* w) {
  if (!w) return;
  msbb_system* s = w->owner;
  w->w.reset();
  delete w;
  system_unref(s);
}
int32_t msbb_prove(msbb_system* sys, msbb_witness* w, uint8_t* proof_out, size_t cap, size_t* proof_len, double* stage_ms) {
  BB_TRY
  if (!sys || !w || !proof_len) throw std::runtime_error("null argument");
  if (w->w->sys != sys->sys.get()) throw std::runtime_error("witness belongs to another system");
  std::vector<uint8_t> bytes = prove(*sys->sys, *w->w, stage_ms);
  *proof_len = bytes.size();
  if (bytes.size() > cap || !proof_out) return MS_ERR_BUFFER;
  memcpy(proof_out, bytes.data(), bytes.size());
  return MS_OK;
  BB_CATCH
}

int32_t msbb_verify(msbb_system* sys, size_t n_claims, const uint64_t* claim_offsets, const uint32_t* claim_data, const uint8_t* proof,
                    size_t proof_len, int32_t* verdict) {
  BB_TRY
  if (!sys || !verdict || (!proof && proof_len) || (n_claims && (!claim_offsets || !claim_data))) throw std::runtime_error("null argument");
  static const uint64_t zero = 0;
  *verdict = verify(*sys->sys, n_claims, n_claims ? claim_offsets : &zero, claim_data, proof, proof_len);
  return MS_OK;
  BB_CATCH
}

// ---- PCS-level entry points
int32_t msbb_set_poseidon2(ms_ctx* ctx, const uint32_t* k141) {
  BB_TRY
  Ctx& c = *msamd::ctx_of(ctx);
  HIP_CHECK(hipSetDevice(c.device));
  PermHolder& h = perm_holder(c.device);
  for (int i = 0; i < 141; i++)
    if (k141[i] >= BB_P) throw std::runtime_error("non-canonical round constant");
  for (int r = 0; r < 8; r++)
    for (int i = 0; i < 16; i++) h.host.external[r][i] = bb_to_monty(k141[16 * r + i]);
  for (int r = 0; r < 13; r++) h.host.internal[r] = bb_to_monty(k141[128 + r]);
  set_internal_diag(h.host);
  if (!h.dev) HIP_CHECK(hipMalloc(&h.dev, sizeof(Poseidon2)));
  c.h2d(h.dev, &h.host, sizeof(Poseidon2));
  c.sync();
  h.set = true;
  return MS_OK;
  BB_CATCH
}
static const Poseidon2* need_perm(Ctx& c) {
  PermHolder& h = perm_holder(c.device);
  if (!h.set) throw std::runtime_error("msbb_set_poseidon2 has not been called for this device");
  return h.dev;
}
int32_t msbb_poseidon2_permute(ms_ctx* ctx, uint32_t* states, size_t n) {
  BB_TRY
  Ctx& c = *msamd::ctx_of(ctx);
  HIP_CHECK(hipSetDevice(c.device));
  const Poseidon2* perm = need_perm(c);
  if (!n) return MS_OK;
  std::vector<u32> m(16 * n);
  for (size_t i = 0; i < 16 * n; i++) {
    if (states[i] >= BB_P) throw std::runtime_error("non-canonical state word");
    m[i] = bb_to_monty(states[i]);
  }
  DBuf<u32> d(c, 16 * n);
  c.h2d(d.p, m.data(), m.size() * 4);
  bb_permute_batch(c, perm, d.p, n);
  c.d2h(m.data(), d.p, m.size() * 4);
  for (size_t i = 0; i < 16 * n; i++) states[i] = bb_from_monty(m[i]);
  return MS_OK;
  BB_CATCH
}
int32_t msbb_dft_batch(ms_ctx* ctx, const uint32_t* in, size_t h, size_t w, int32_t inverse, uint32_t* out) {
  BB_TRY
  Ctx& c = *msamd::ctx_of(ctx);
  HIP_CHECK(hipSetDevice(c.device));
  if (h == 0 || (h & (h - 1))) throw std::runtime_error("height must be a power of two");
  if (!w) return MS_OK;
  unsigned log_h = log2_strict(h);
  BMat m;
  bb_upload_rows(c, in, h, w, m);
  bb_dif(c, m.buf.p, m.ld, log_h, w);
  std::vector<u32> rows(h * w);
  bb_download_rows(c, m, true, rows.data());  // DIF leaves the result bit-reversed
  if (!inverse) {
    memcpy(out, rows.data(), rows.size() * 4);
    return MS_OK;
  }
  // x[k] = X[(h - k) mod h] / h with X the forward transform
  u32 h_inv = bb_inv(bb_to_monty((u32)(h % BB_P)));
  for (size_t k = 0; k < h; k++)
    for (size_t j = 0; j < w; j++) out[k * w + j] = bb_from_monty(bb_mul(bb_to_monty(rows[((h - k) & (h - 1)) * w + j]), h_inv));
  return MS_OK;
  BB_CATCH
}
int32_t msbb_coset_lde_batch(ms_ctx* ctx, const uint32_t* in, size_t h, size_t w, uint32_t log_blowup, uint32_t* out) {
  BB_TRY
  Ctx& c = *msamd::ctx_of(ctx);
  HIP_CHECK(hipSetDevice(c.device));
  if (h == 0 || (h & (h - 1))) throw std::runtime_error("height must be a power of two");
  if (log2_strict(h) + log_blowup > BB_TWO_ADICITY) throw std::runtime_error("LDE taller than the two-adicity of BabyBear");
  if (!w) return MS_OK;
  BMat m, lde;
  bb_upload_rows(c, in, h, w, m);
  bb_coset_lde(c, m, log_blowup, lde);
  bb_download_rows(c, lde, false, out);
  return MS_OK;
  BB_CATCH
}
int32_t msbb_mmcs_commit(ms_ctx* ctx, size_t n, const uint32_t* const* mats, const uint64_t* heights, const uint64_t* widths,
                         uint32_t cap_height, uint32_t* cap_out, msbb_mmcs** out) {
  BB_TRY
  Ctx& c = *msamd::ctx_of(ctx);
  HIP_CHECK(hipSetDevice(c.device));
  const Poseidon2* perm = need_perm(c);
  std::unique_ptr<msbb_mmcs> h(new msbb_mmcs());
  h->ctx = &c;
  std::vector<BMat> ms(n);
  for (size_t i = 0; i < n; i++) bb_upload_rows(c, mats[i], (size_t)heights[i], (size_t)widths[i], ms[i]);
  bb_commit(c, perm, std::move(ms), cap_height, h->data);
  std::vector<Digest8> cap = tree_cap(c, h->data.tree);
  for (size_t i = 0; i < cap.size(); i++)
    for (int k = 0; k < 8; k++) cap_out[8 * i + k] = bb_from_monty(cap[i].w[k]);
  h->owner = ctx;
  msamd::ctx_retain(ctx);
  *out = h.release();
  return MS_OK;
  BB_CATCH
}
int32_t msbb_mmcs_open(msbb_mmcs* m, size_t index, uint32_t* vals_out, uint32_t* proof_out, size_t* n_siblings) {
  BB_TRY
  Ctx& c = *m->ctx;
  HIP_CHECK(hipSetDevice(c.device));
  const BTree& t = m->d().tree;
  unsigned log_max = log2_strict(t.sizes[0]);
  if (index >= t.sizes[0]) throw std::runtime_error("index out of range");
  std::vector<GatherSeg> segs;
  u32 pos = 0;
  for (auto& mat : m->d().ldes) {
    size_t row = index >> (log_max - log2_strict(mat.h));
    segs.push_back(GatherSeg{mat.buf.p + row, pos, (u32)mat.w, (u32)mat.ld});
    pos += (u32)mat.w;
  }
  u32 vals = pos;
  size_t ch_eff = std::min<size_t>(t.cap_height, t.layers.size() - 1);
  for (size_t i = 0; i + ch_eff < log_max; i++) {
    segs.push_back(GatherSeg{(const u32*)(t.layers[i].p + ((index >> i) ^ 1)), pos, 8, 1});
    pos += 8;
  }
  std::vector<u32> g;
  bb_gather(c, segs, g);
  for (u32 k = 0; k < vals; k++) vals_out[k] = bb_from_monty(g[k]);
  for (u32 k = vals; k < pos; k++) proof_out[k - vals] = bb_from_monty(g[k]);
  *n_siblings = log_max - ch_eff;
  return MS_OK;
  BB_CATCH
}
void msbb_mmcs_destroy(msbb_mmcs* m) {
  if (!m) return;
  ms_ctx* c = m->owner;
  msbb_system* s = m->sys_ref;
  m->data = BPcsData();
  delete m;
  if (s) system_unref(s);
  msamd::ctx_release(c);
}
}  // extern "C"
// ---- Level 2 (include/mstark_bb.h): the steps of prove() above, one call each, on device handles
static E4 e4_in(const uint32_t w[4]) {
  E4 e;
  for (int k = 0; k < 4; k++) {
    if (w[k] >= BB_P) throw std::runtime_error("non-canonical extension-field coordinate");
    e.c[k] = bb_to_monty(w[k]);
  }
  return e;
}
static void e4_out(E4 e, uint32_t w[4]) {
  for (int k = 0; k < 4; k++) w[k] = bb_from_monty(e.c[k]);
}
static void cap_out_words(const std::vector<Digest8>& cap, uint32_t* out) {
  for (size_t i = 0; i < cap.size(); i++)
    for (int k = 0; k < 8; k++) out[8 * i + k] = bb_from_monty(cap[i].w[k]);
}
static msbb_mmcs* new_mmcs(msbb_system* sys) {
  msbb_mmcs* h = new msbb_mmcs();
  h->ctx = sys->sys->ctx;
  h->owner = sys->owner;
  msamd::ctx_retain(sys->owner);
  return h;
}
static msbb_trace* new_trace(msbb_system* sys, BMat&& m, int kind) {
  msbb_trace* t = new msbb_trace();
  t->m = std::move(m);
  t->kind = kind;
  t->owner = sys->owner;
  msamd::ctx_retain(sys->owner);
  return t;
}
static std::vector<size_t> active_circuits(const BWitness& w) {
  std::vector<size_t> a;
  for (size_t i = 0; i < w.heights.size(); i++)
    if (w.heights[i]) a.push_back(i);
  return a;
}
static void need_device_witness(const msbb_witness* w) {
  if (!w) throw std::runtime_error("null argument");
  if (w->w->host_resident) throw std::runtime_error("the step-wise entry points need a device-resident witness (msbb_witness_create)");
}
typedef std::unique_ptr<msbb_mmcs, void (*)(msbb_mmcs*)> MmcsPtr;
extern "C" {

int32_t msbb_challenger_create(msbb_system* sys, msbb_challenger** out) {
  BB_TRY
  if (!sys || !out) throw std::runtime_error("null argument");
  std::unique_ptr<msbb_challenger> h(new msbb_challenger());
  h->ch.reset(new Challenger(&sys->sys->perm));
  for (u32 v : sys->sys->seed) h->ch->observe(v);  // config.initialise_challenger(), baby_bear_config.rs:108-114
  h->owner = sys;
  sys->refs++;
  *out = h.release();
  return MS_OK;
  BB_CATCH
}
void msbb_challenger_destroy(msbb_challenger* ch) {
  if (!ch) return;
  msbb_system* s = ch->owner;
  delete ch;
  system_unref(s);
}
int32_t msbb_challenger_observe(msbb_challenger* ch, const uint32_t* elems, size_t n) {
  BB_TRY
  if (!ch || (n && !elems)) throw std::runtime_error("null argument");
  for (size_t i = 0; i < n; i++)
    if (elems[i] >= BB_P) throw std::runtime_error("non-canonical value");
  for (size_t i = 0; i < n; i++) ch->ch->observe(bb_to_monty(elems[i]));
  return MS_OK;
  BB_CATCH
}
int32_t msbb_challenger_observe_digests(msbb_challenger* ch, const uint32_t* digests, size_t n) {
  return msbb_challenger_observe(ch, digests, 8 * n);
}
int32_t msbb_challenger_sample_ext(msbb_challenger* ch, uint32_t out4[4]) {
  BB_TRY
  if (!ch || !out4) throw std::runtime_error("null argument");
  e4_out(ch->ch->sample_e4(), out4);
  return MS_OK;
  BB_CATCH
}
int32_t msbb_challenger_sample_bits(msbb_challenger* ch, uint32_t bits, uint64_t* out) {
  BB_TRY
  if (!ch || !out || bits > 30) throw std::runtime_error("bad argument");
  *out = ch->ch->sample_bits(bits);
  return MS_OK;
  BB_CATCH
}
int32_t msbb_challenger_observe_claims(msbb_challenger* ch, msbb_witness* w) {
  BB_TRY
  if (!ch || !w) throw std::runtime_error("null argument");
  Challenger& c = *ch->ch;
  c.observe_usize(w->w->claims.size());  // src/prover.rs:369-373
  for (auto& cl : w->w->claims) {
    c.observe_usize(cl.size());
    for (u32 x : cl) c.observe(bb_to_monty(x));
  }
  return MS_OK;
  BB_CATCH
}
void msbb_trace_destroy(msbb_trace* t) {
  if (!t) return;
  ms_ctx* c = t->owner;
  t->m = BMat();
  delete t;
  msamd::ctx_release(c);
}
int32_t msbb_trace_info(const msbb_trace* t, uint64_t out3[3]) {
  if (!t || !out3) return MS_ERR;
  out3[0] = t->m.h, out3[1] = t->m.w, out3[2] = (uint64_t)t->kind;
  return MS_OK;
}
int32_t msbb_system_preprocessed_mmcs(msbb_system* sys, msbb_mmcs** out) {
  BB_TRY
  if (!sys || !out) throw std::runtime_error("null argument");
  *out = nullptr;
  if (!sys->sys->has_pre) return MS_OK;
  msbb_mmcs* h = new_mmcs(sys);
  h->view = &sys->sys->pre_data;
  h->sys_ref = sys;
  sys->refs++;
  *out = h;
  return MS_OK;
  BB_CATCH
}
int32_t msbb_witness_commit_stage1(msbb_witness* w, uint32_t* cap_out, msbb_mmcs** out) {
  BB_TRY
  need_device_witness(w);
  if (!cap_out || !out) throw std::runtime_error("null argument");
  BSystem& sys = *w->w->sys;
  Ctx& ctx = *sys.ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  std::vector<BMat> ldes;
  for (size_t ci : active_circuits(*w->w)) {
    ldes.emplace_back();
    bb_coset_lde(ctx, w->w->traces[ci], (unsigned)sys.params.log_blowup, ldes.back());
  }
  if (ldes.empty()) throw std::runtime_error("cannot prove with every circuit deactivated");
  MmcsPtr h(new_mmcs(w->owner), msbb_mmcs_destroy);
  bb_commit(ctx, sys.d_perm.p, std::move(ldes), (unsigned)sys.params.cap_height, h->data);
  cap_out_words(tree_cap(ctx, h->data.tree), cap_out);
  *out = h.release();
  return MS_OK;
  BB_CATCH
}
int32_t msbb_witness_claims_accumulator(msbb_witness* w, const uint32_t beta[4], const uint32_t gamma[4], uint32_t acc_out[4]) {
  BB_TRY
  need_device_witness(w);
  BSystem& sys = *w->w->sys;
  HIP_CHECK(hipSetDevice(sys.ctx->device));
  e4_out(bb_claims_accumulator(*sys.ctx, w->w->d_claim_data.p, w->w->d_claim_offs.p, w->w->claims.size(), e4_in(beta), e4_in(gamma)), acc_out);
  return MS_OK;
  BB_CATCH
}
int32_t msbb_stage2_build(msbb_witness* w, const uint32_t beta[4], const uint32_t gamma[4], const uint32_t acc_in[4], uint32_t* accs_out,
                          msbb_trace** traces_out) {
  BB_TRY
  need_device_witness(w);
  if (!accs_out || !traces_out) throw std::runtime_error("null argument");
  BSystem& sys = *w->w->sys;
  Ctx& ctx = *sys.ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  const E4 b = e4_in(beta), g = e4_in(gamma);
  E4 running = e4_in(acc_in);
  std::vector<std::unique_ptr<msbb_trace, void (*)(msbb_trace*)>> made;
  size_t pos = 0;
  for (size_t ci : active_circuits(*w->w)) {
    const BCircuit& c = sys.circuits[ci];
    BMat t;
    E4 total;
    bb_stage2(ctx, c.prog, c.lookup_prefix_len, c.lk, w->w->traces[ci], c.pre_width ? &c.pre : nullptr, b, g, t, &total);
    running = e4_add(running, total);  // src/lookup.rs:544-550
    e4_out(running, accs_out + 4 * pos);
    made.emplace_back(new_trace(w->owner, std::move(t), 0), msbb_trace_destroy);
    pos++;
  }
  for (size_t i = 0; i < made.size(); i++) traces_out[i] = made[i].release();
  return MS_OK;
  BB_CATCH
}
static int32_t commit_handles(msbb_system* sys, size_t n, msbb_trace* const* ts, int kind, uint32_t* cap_out, msbb_mmcs** out) {
  BB_TRY
  if (!sys || !n || !ts || !cap_out || !out) throw std::runtime_error("null argument");
  BSystem& s = *sys->sys;
  Ctx& ctx = *s.ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  for (size_t i = 0; i < n; i++) {
    if (!ts[i] || ts[i]->kind != kind || !ts[i]->m.h) throw std::runtime_error(kind ? "expected an LDE handle" : "expected an evaluation handle (already consumed?)");
    if (ts[i]->owner != sys->owner) throw std::runtime_error("handle belongs to another context");
  }
  std::vector<BMat> ldes;
  for (size_t i = 0; i < n; i++) {
    if (kind) {
      ldes.push_back(std::move(ts[i]->m));
    } else {
      ldes.emplace_back();
      bb_coset_lde(ctx, ts[i]->m, (unsigned)s.params.log_blowup, ldes.back());
    }
    ts[i]->m = BMat();  // consumed
  }
  MmcsPtr h(new_mmcs(sys), msbb_mmcs_destroy);
  bb_commit(ctx, s.d_perm.p, std::move(ldes), (unsigned)s.params.cap_height, h->data);
  cap_out_words(tree_cap(ctx, h->data.tree), cap_out);
  *out = h.release();
  return MS_OK;
  BB_CATCH
}
int32_t msbb_pcs_commit_traces(msbb_system* sys, size_t n, msbb_trace* const* evals, uint32_t* cap_out, msbb_mmcs** out) {
  return commit_handles(sys, n, evals, 0, cap_out, out);
}
int32_t msbb_pcs_commit_ldes(msbb_system* sys, size_t n, msbb_trace* const* ldes, uint32_t* cap_out, msbb_mmcs** out) {
  return commit_handles(sys, n, ldes, 1, cap_out, out);
}
int32_t msbb_quotient(msbb_system* sys, size_t circuit, uint32_t log_n, msbb_mmcs* s1, size_t s1_idx, msbb_mmcs* s2, size_t s2_idx,
                      const uint32_t publics16[16], const uint32_t alpha[4], msbb_trace** q_lde_out) {
  BB_TRY
  if (!sys || !s1 || !s2 || !publics16 || !alpha || !q_lde_out) throw std::runtime_error("null argument");
  BSystem& s = *sys->sys;
  Ctx& ctx = *s.ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  if (circuit >= s.circuits.size()) throw std::runtime_error("circuit index out of range");
  if (s1_idx >= s1->d().ldes.size() || s2_idx >= s2->d().ldes.size()) throw std::runtime_error("matrix index out of range");
  const BCircuit& c = s.circuits[circuit];
  const unsigned lb = (unsigned)s.params.log_blowup, log_q = log2_strict(c.quotient_degree());
  const BMat &m1 = s1->d().ldes[s1_idx], &m2 = s2->d().ldes[s2_idx];
  if (m1.h != (size_t(1) << (log_n + lb)) || m2.h != m1.h || m1.w != c.main_width || m2.w != c.stage2_width)
    throw std::runtime_error("the committed matrices do not have this circuit's shape");
  BQuotientIn in;
  in.prog = &c.prog, in.lk = &c.lk, in.d_zeros = c.d_zeros.p, in.n_zeros = c.zeros.size(), in.constraint_count = c.constraint_count;
  in.jit = &c.quotient_jit;
  in.pre = s.has_pre && s.pre_indices[circuit] >= 0 ? &s.pre_data.ldes[s.pre_indices[circuit]] : nullptr;
  in.s1 = &m1, in.s2 = &m2;
  in.log_n = log_n, in.log_q = log_q, in.log_blowup = lb;
  for (int k = 0; k < 4; k++) in.publics[k] = e4_in(publics16 + 4 * k);  // beta, gamma, acc_in, acc_out (src/lookup.rs:78-84)
  in.alpha = e4_in(alpha);
  BMat q_evals, q_lde;
  bb_quotient(ctx, in, q_evals);
  bb_quotient_lde(ctx, q_evals, log_n, log_q, lb, q_lde);
  *q_lde_out = new_trace(sys, std::move(q_lde), 1);
  return MS_OK;
  BB_CATCH
}
int32_t msbb_pcs_open(msbb_system* sys, size_t n_rounds, msbb_mmcs* const* rounds, const uint64_t* n_points, const uint32_t* points,
                      msbb_challenger* ch, uint32_t* opened_out, size_t opened_cap_words, uint8_t* fri_out, size_t fri_cap, size_t* fri_len) {
  BB_TRY
  if (!sys || !n_rounds || !rounds || !n_points || !ch || !fri_len) throw std::runtime_error("null argument");
  if (ch->owner != sys) throw std::runtime_error("challenger belongs to another system");
  BSystem& s = *sys->sys;
  HIP_CHECK(hipSetDevice(s.ctx->device));
  std::vector<OpenRound> rs(n_rounds);
  size_t mi = 0, pi = 0, need_words = 0;
  for (size_t r = 0; r < n_rounds; r++) {
    if (!rounds[r] || rounds[r]->owner != sys->owner) throw std::runtime_error("round handle missing or of another context");
    rs[r].data = &rounds[r]->d();
    for (auto& m : rs[r].data->ldes) {
      const size_t np = (size_t)n_points[mi++];
      if (np > 2) throw std::runtime_error("at most two opening points per matrix");
      std::vector<E4> pts;
      for (size_t k = 0; k < np; k++) pts.push_back(e4_in(points + 4 * (pi++)));
      need_words += np * m.w * 4;
      rs[r].points.push_back(std::move(pts));
    }
  }
  std::vector<OpenedRound> opened;
  FriOut fri;
  pcs_open(s, rs, *ch->ch, opened, fri);
  W w;
  write_fri(w, fri, s.params);
  *fri_len = w.b.size();
  if (need_words > opened_cap_words || w.b.size() > fri_cap || !opened_out || !fri_out) return MS_ERR_BUFFER;
  uint32_t* o = opened_out;
  for (auto& r : opened)
    for (auto& m : r)
      for (auto& pt : m)
        for (auto& e : pt) {
          e4_out(e, o);
          o += 4;
        }
  memcpy(fri_out, w.b.data(), w.b.size());
  return MS_OK;
  BB_CATCH
}
int32_t msbb_field_op(ms_ctx* ctx, int32_t op, const uint32_t* a, const uint32_t* b, size_t n, uint32_t* out) {
  BB_TRY
  if (op < 0 || op > 5) throw std::runtime_error("bad op");
  bb_field_op(*msamd::ctx_of(ctx), op, a, b, n, out);
  return MS_OK;
  BB_CATCH
}

}  // extern "C"
