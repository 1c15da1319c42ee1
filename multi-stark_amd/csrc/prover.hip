// Host orchestration of the MI355X prover: System::new, SystemWitness, prove_multiple_claims, PCS open / FRI.
// Stage order, transcript order and proof container layout follow /root/reference/src/prover.rs:290-603
// (observe/sample sequence :297-431,527,539; rounds :540-579; Proof fields :213-238) and src/system.rs:115-222.
// Plonky3-side behaviour (TwoAdicFriPcs::open, prove_fri, challenger, MMCS) follows the published p3 0.5.1
// algorithms; the LDE matrices never leave the device — only caps, opened values, the final polynomial and the
// query openings cross PCIe.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <sched.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "host.h"
#include "lookup_params.h"
#include "quotient_params.h"

namespace msamd {

// ------------------------------------------------------------------ host BLAKE3 + challenger
namespace {
void load_block(const uint8_t* p, size_t len, u32 w[16]) {
  uint8_t buf[64];
  memset(buf, 0, 64);
  if (len) memcpy(buf, p, len);
  for (int i = 0; i < 16; i++)
    w[i] = (u32)buf[4 * i] | ((u32)buf[4 * i + 1] << 8) | ((u32)buf[4 * i + 2] << 16) | ((u32)buf[4 * i + 3] << 24);
}
void subtree_cv(const uint8_t* in, size_t len, u64 chunk_index, u32 root_flag, u32 out[8]) {
  if (len <= 1024) {
    b3_iv(out);
    size_t nblocks = len == 0 ? 1 : (len + 63) / 64;
    for (size_t b = 0; b < nblocks; b++) {
      size_t off = b * 64, bl = len - off < 64 ? len - off : 64;
      u32 m[16];
      load_block(in + off, bl, m);
      u32 flags = (b == 0 ? B3_CHUNK_START : 0) | (b == nblocks - 1 ? (B3_CHUNK_END | root_flag) : 0);
      b3_compress(out, m, chunk_index, (u32)bl, flags);
    }
    return;
  }
  size_t full = (len - 1) / 1024, lc = 1;
  while (lc * 2 <= full) lc *= 2;
  u32 m[16];
  subtree_cv(in, lc * 1024, chunk_index, 0, m);
  subtree_cv(in + lc * 1024, len - lc * 1024, chunk_index + lc, 0, m + 8);
  b3_iv(out);
  b3_compress(out, m, 0, 64, B3_PARENT | root_flag);
}
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// MSAMD_TRACE_HOST=1: host-side time stamps of one proof (where the host is when; printed when prove() returns)
struct HostProbes {
  bool on = false;
  double t0 = 0;
  long faults0 = 0;
  struct Mark {
    const char* what;
    double t;
    long faults;  // minor page faults of this thread so far
  };
  std::vector<Mark> marks;
  static long faults_now() {
    struct rusage ru;
    return getrusage(RUSAGE_THREAD, &ru) == 0 ? ru.ru_minflt : 0;
  }
  void start() {
    on = getenv("MSAMD_TRACE_HOST") != nullptr;
    marks.clear();
    t0 = now_ms();
    if (on) faults0 = faults_now();
  }
  void mark(const char* what) {
    if (on) marks.push_back(Mark{what, now_ms(), faults_now()});
  }
  void print() const {
    if (!on) return;
    double prev = t0;
    long pf = faults0;
    for (auto& m : marks) {
      fprintf(stderr, "[msamd]   %-34s at %8.1f us (+%7.1f)  page faults +%ld\n", m.what, 1e3 * (m.t - t0), 1e3 * (m.t - prev), m.faults - pf);
      prev = m.t;
      pf = m.faults;
    }
  }
};
thread_local HostProbes g_probes;
}  // namespace

void blake3_host(const uint8_t* in, size_t len, uint8_t out[32]) {
  u32 cv[8];
  subtree_cv(in, len, 0, B3_ROOT, cv);
  for (int i = 0; i < 8; i++)
    for (int k = 0; k < 4; k++) out[4 * i + k] = (uint8_t)(cv[i] >> (8 * k));
}

uint8_t Challenger::sample_byte() {
  if (output.empty()) {
    uint8_t d[32];
    blake3_host(input.data(), input.size(), d);
    input.assign(d, d + 32);
    output.assign(d, d + 32);
  }
  uint8_t b = output.back();
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
    if (v < GL_P) return v;
  }
}
E2 Challenger::sample_ext() {
  u64 a = sample_base();
  u64 b = sample_base();
  return e2(a, b);
}
size_t Challenger::sample_bits(unsigned bits) { return (size_t)(sample_u64() & ((u64(1) << bits) - 1)); }

// smallest witness w such that observing w then sampling `bits` bits gives zero (SURVEY finding 7);
// ZERO at 0 bits (DeterministicPow, src/types.rs:75-80). The candidate transcript is input || w (8 bytes).
u64 Challenger::grind(unsigned bits) {
  if (bits == 0) return 0;
  std::vector<uint8_t> buf(input);
  size_t base = buf.size();
  buf.resize(base + 8);
  const u64 mask = (u64(1) << bits) - 1;
  for (u64 w = 0;; w++) {
    for (int k = 0; k < 8; k++) buf[base + k] = (uint8_t)(w >> (8 * k));
    uint8_t d[32];
    blake3_host(buf.data(), buf.size(), d);
    // sample_bits pops 8 bytes from the back of the digest: byte k of the u64 is d[31 - k]
    u64 v = 0;
    for (int k = 0; k < 8; k++) v |= (u64)d[31 - k] << (8 * k);
    if ((v & mask) == 0) {
      observe(w);
      size_t s = sample_bits(bits);
      if (s != 0) throw std::runtime_error("grind: internal inconsistency");
      return w;
    }
  }
}

// ------------------------------------------------------------------ System::new
namespace {
struct Reader {
  const uint8_t* p;
  size_t n, off = 0;
  u64 word() {
    if (off + 8 > n) throw std::runtime_error("system blob truncated");
    u64 v;
    memcpy(&v, p + off, 8);
    off += 8;
    return v;
  }
};
const u64 BLOB_MAGIC = 0x31305359534D0000ULL;
}  // namespace

void commit_matrices(Ctx& ctx, std::vector<DMat>&& ldes, unsigned cap_height, PcsData& out) {
  out.ldes = std::move(ldes);
  out.tree = DTree();
  out.tree.cap_height = cap_height;
  for (auto& m : out.ldes) {
    out.tree.mat_d.push_back(m.d());
    out.tree.mat_h.push_back(m.h);
    out.tree.mat_w.push_back(m.w);
  }
  merkle_build(ctx, out.tree);
}

// host row-major evaluations -> device bit-reversed coset LDE
// (`early`: the matrix already transposed and through the first pass of the inverse transform - HostUpload's row groups; taken over)
static DMat lde_of_host_matrix(Ctx& ctx, const u64* rowmajor_dev, size_t h, size_t w, unsigned lb, DBuf<u64>* early = nullptr) {
  unsigned logn = log2_strict(h);
  const bool first_pass_done = early && early->p;
  DBuf<u64> ev = first_pass_done ? std::move(*early) : DBuf<u64>(ctx, h * w);
  if (!first_pass_done) transpose_in(ctx, rowmajor_dev, ev.p, h, w, true);
  DMat lde;
  lde.h = h << lb;
  lde.w = w;
  lde.buf = DBuf<u64>(ctx, lde.h * w);
  coset_lde(ctx, ev.p, lde.d(), logn, lb, w, first_pass_done);
  return lde;
}

std::unique_ptr<HSystem> system_from_blob(Ctx& ctx, const uint8_t* blob, size_t len) {
  HIP_CHECK(hipSetDevice(ctx.device));  // kernels, modules and pool blocks of this system belong to the context's device
  Reader rd{blob, len};
  if (rd.word() != BLOB_MAGIC) throw std::runtime_error("bad system blob magic");
  std::unique_ptr<HSystem> sys(new HSystem());
  sys->ctx = &ctx;
  Params& p = sys->params;
  p.log_blowup = rd.word();
  p.cap_height = rd.word();
  p.log_final_poly_len = rd.word();
  p.max_log_arity = rd.word();
  p.num_queries = rd.word();
  p.commit_pow_bits = rd.word();
  p.query_pow_bits = rd.word();
  if (p.max_log_arity < 1 || p.max_log_arity > FRI_MAX_LOG_ARITY) throw std::runtime_error("max_log_arity must be 1 .. 6 (a FRI row is hashed as one BLAKE3 chunk)");
  if (p.log_blowup < 1 || p.log_blowup > 8) throw std::runtime_error("log_blowup out of range");
  if (p.commit_pow_bits > 40 || p.query_pow_bits > 40) throw std::runtime_error("proof-of-work bits out of range");
  {
    const char* tag = "multi-stark/v0";  // src/types.rs:118-130
    sys->seed.assign(tag, tag + 14);
    const u64 ps[7] = {p.log_blowup, p.cap_height, p.log_final_poly_len, p.max_log_arity, p.num_queries, p.commit_pow_bits,
                       p.query_pow_bits};
    for (u64 x : ps)
      for (int k = 0; k < 8; k++) sys->seed.push_back((uint8_t)(x >> (8 * k)));
  }
  const size_t D = 2;
  size_t nc = rd.word();
  std::vector<DMat> pre_ldes;
  for (size_t ci = 0; ci < nc; ci++) {
    sys->circuits.emplace_back();
    HCircuit& c = sys->circuits.back();
    c.main_width = rd.word();
    c.pre_width = rd.word();
    c.pre_height = rd.word();
    size_t nn = rd.word(), nz = rd.word(), nl = rd.word();
    c.num_lookups = nl;
    c.stage2_width = std::max<size_t>(nl, 1) * D;
    c.nodes.resize(nn);
    c.degrees.resize(nn);
    for (size_t i = 0; i < nn; i++) {
      u64 w0 = rd.word();
      PNode& nd = c.nodes[i];
      nd.kind = (uint32_t)(w0 & 0xff);
      nd.source = (uint32_t)((w0 >> 8) & 0xff);
      nd.offset = (uint32_t)((w0 >> 16) & 0xff);
      nd.a = rd.word();
      nd.b = rd.word();
      auto child = [&](u64 id) -> uint32_t {
        if (id >= i) throw std::runtime_error("node program is not topologically ordered");
        return c.degrees[id];
      };
      uint32_t deg = 0;
      switch (nd.kind) {  // src/graph.rs:242-252
        case OP_CONST:
          if (nd.a >= GL_P) throw std::runtime_error("non-canonical constant in node program");
          break;
        case OP_PUBLIC:
          if (nd.a >= 4 * D) throw std::runtime_error("public index out of range");
          break;
        case OP_IS_TRANS: break;
        case OP_VAR: {
          size_t width = nd.source == 0 ? c.pre_width : nd.source == 1 ? c.main_width : c.stage2_width;
          if (nd.source > 2 || nd.offset > 1 || nd.a >= width) throw std::runtime_error("column reference out of range");
          deg = 1;
          break;
        }
        case OP_IS_FIRST:
        case OP_IS_LAST: deg = 1; break;
        case OP_ADD:
        case OP_SUB: deg = std::max(child(nd.a), child(nd.b)); break;
        case OP_MUL: deg = child(nd.a) + child(nd.b); break;
        case OP_NEG: deg = child(nd.a); break;
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
        msg = std::max(msg, c.degrees[a]);
        c.lookup_prefix_len = std::max<size_t>(c.lookup_prefix_len, a + 1);
      }
      c.args_width += na;
      logup_deg = std::max(logup_deg, std::max(msg + 1, c.degrees[m]));
      c.lookups.emplace_back((uint32_t)m, std::move(args));
    }
    c.constraint_count = nz + std::max<size_t>(nl, 1) * D;  // src/system.rs:151
    c.max_constraint_degree = std::max(graph_deg, logup_deg);
    if (c.quotient_degree() > (size_t(1) << p.log_blowup))  // src/system.rs:171-178
      throw std::runtime_error("circuit " + std::to_string(ci) + ": constraint degree needs a quotient degree beyond the blowup");
    build_program(ctx, c.nodes, c.zeros, c.lookups, c.prog);
    quotient_jit_build(c.nodes, c.zeros, c.lookups, c.quotient_degree(), c.prog.jit);  // the circuit's own kernel, from hiprtc or the cache
    {
      std::vector<uint32_t> counts;
      for (auto& l : c.lookups) counts.push_back((uint32_t)l.second.size());
      stage2_jit_build(counts, c.stage2_jit);
      stage2_trace_jit_build(c.nodes, c.lookups, c.main_width, c.pre_width, c.lookup_prefix_len, c.stage2_trace_jit);
    }
    {
      // the lookup prefix may only read trace columns and row selectors (src/graph.rs: Stage2InBaseContext; publics
      // do not exist at witness time); anything else keeps the host sweep, which reports the error
      bool ok = !c.lookups.empty();
      for (size_t i = 0; i < c.lookup_prefix_len && ok; i++)
        ok = !(c.nodes[i].kind == OP_PUBLIC || (c.nodes[i].kind == OP_VAR && c.nodes[i].source == 2));
      if (ok) {
        build_program(ctx, c.nodes, std::vector<uint32_t>(), c.lookups, c.prefix_prog);
        c.prefix_on_device = true;
      }
    }
    c.prog.constraint_count = c.constraint_count;
    c.prog.main_w = c.main_width;
    c.prog.pre_w = c.pre_width;
    c.prog.s2_w = c.stage2_width;
    if (c.pre_width) {
      if (c.pre_height == 0 || (c.pre_height & (c.pre_height - 1))) throw std::runtime_error("preprocessed height must be a power of two");
      if (log2_strict(c.pre_height) > NTT_MAX_LOG) throw std::runtime_error("preprocessed trace too tall");
      size_t cnt = c.pre_height * c.pre_width;
      c.preprocessed.resize(cnt);
      for (auto& x : c.preprocessed) {
        x = rd.word();
        if (x >= GL_P) throw std::runtime_error("non-canonical preprocessed value");
      }
      c.d_preprocessed = DBuf<u64>(ctx, cnt);
      ctx.h2d(c.d_preprocessed.p, c.preprocessed.data(), cnt * 8);
      sys->pre_indices.push_back((int)pre_ldes.size());
      pre_ldes.push_back(lde_of_host_matrix(ctx, c.d_preprocessed.p, c.pre_height, c.pre_width, (unsigned)p.log_blowup));
      ctx.sync();
    } else {
      c.pre_height = 0;
      sys->pre_indices.push_back(-1);
    }
  }
  if (rd.off != len) throw std::runtime_error("trailing bytes in system blob");
  if (!pre_ldes.empty()) {
    sys->has_pre = true;
    commit_matrices(ctx, std::move(pre_ldes), (unsigned)p.cap_height, sys->pre_data);
    sys->pre_commit = merkle_cap(ctx, sys->pre_data.tree);
  }
  ctx.sync();
  return sys;
}

// ------------------------------------------------------------------ SystemWitness
namespace {
// host sweep of the lookup prefix for one row (src/eval.rs:59-106 in the base field)
void sweep_prefix(const HCircuit& c, const u64* pre_cur, const u64* pre_next, const u64* cur, const u64* next, bool first,
                  bool last, std::vector<u64>& buf) {
  size_t len = c.lookup_prefix_len;
  buf.resize(len);
  for (size_t i = 0; i < len; i++) {
    const PNode& n = c.nodes[i];
    u64 v = 0;
    switch (n.kind) {
      case OP_CONST: v = n.a; break;
      case OP_VAR:
        if (n.source == 0)
          v = (n.offset ? pre_next : pre_cur)[n.a];
        else if (n.source == 1)
          v = (n.offset ? next : cur)[n.a];
        else
          throw std::runtime_error("stage-2 column in a lookup expression");
        break;
      case OP_PUBLIC: throw std::runtime_error("public input in a lookup expression");
      case OP_IS_FIRST: v = first; break;
      case OP_IS_LAST: v = last; break;
      case OP_IS_TRANS: v = !last; break;
      case OP_ADD: v = gl_add(buf[n.a], buf[n.b]); break;
      case OP_SUB: v = gl_sub(buf[n.a], buf[n.b]); break;
      case OP_MUL: v = gl_mul(buf[n.a], buf[n.b]); break;
      default: v = gl_neg(buf[n.a]); break;
    }
    buf[i] = v;
  }
}
// SystemWitness::from_stage_1 on the host (src/system.rs:275-328): the flat LookupValues storage of one circuit
void host_lookup_values(const HCircuit& c, const u64* tr, size_t h, std::vector<u64>& hm, std::vector<u64>& ha) {
  std::vector<uint32_t> offs(1, 0);
  for (auto& l : c.lookups) offs.push_back(offs.back() + (uint32_t)l.second.size());
  hm.assign(h * c.num_lookups, 0);
  ha.assign(h * c.args_width, 0);
  std::vector<u64> buf;
  for (size_t r = 0; r < h; r++) {
    size_t rn = (r + 1) % h;
    const u64* pc = c.pre_width ? &c.preprocessed[r * c.pre_width] : nullptr;
    const u64* pn = c.pre_width ? &c.preprocessed[rn * c.pre_width] : nullptr;
    sweep_prefix(c, pc, pn, tr + r * c.main_width, tr + rn * c.main_width, r == 0, r == h - 1, buf);
    for (size_t j = 0; j < c.num_lookups; j++) {
      hm[r * c.num_lookups + j] = buf[c.lookups[j].first];
      for (size_t k = 0; k < c.lookups[j].second.size(); k++) ha[r * c.args_width + offs[j] + k] = buf[c.lookups[j].second[k]];
    }
  }
}
}  // namespace

// A witness whose traces and claims are already in HBM (witness_gen.hip): SystemWitness::from_stage_1 runs on the device.
std::unique_ptr<HWitness> witness_from_device(HSystem& sys, std::vector<DBuf<u64>>&& traces, const std::vector<size_t>& heights,
                                              DBuf<u64>&& d_claim_offsets, DBuf<u64>&& d_claim_data, size_t n_claims, size_t claim_elems) {
  Ctx& ctx = *sys.ctx;
  const size_t C = sys.circuits.size();
  if (traces.size() != C || heights.size() != C) throw std::runtime_error("expected one trace per circuit");
  std::unique_ptr<HWitness> w(new HWitness());
  w->sys = &sys;
  w->heights = heights;
  w->traces = std::move(traces);
  w->lookups.resize(C);
  for (size_t ci = 0; ci < C; ci++) {
    const HCircuit& c = sys.circuits[ci];
    const size_t h = heights[ci];
    if (h == 0) continue;
    if (h & (h - 1)) throw std::runtime_error("trace height must be a power of two");
    if (log2_strict(h) > NTT_MAX_LOG || log2_strict(h) + sys.params.log_blowup > TW_LOG)
      throw std::runtime_error("trace height exceeds the supported maximum");
    if (c.pre_width && h != c.pre_height) throw std::runtime_error("main trace height must equal preprocessed trace height");
    DLookups& lk = w->lookups[ci];
    lk.height = h;
    lk.num_lookups = c.num_lookups;
    lk.args_width = c.args_width;
    if (c.num_lookups == 0) continue;
    std::vector<uint32_t> offs(1, 0);
    for (auto& l : c.lookups) offs.push_back(offs.back() + (uint32_t)l.second.size());
    lk.arg_offsets = DBuf<uint32_t>(ctx, offs.size());
    ctx.h2d(lk.arg_offsets.p, offs.data(), offs.size() * 4);
    if (c.stage2_trace_jit.function && !getenv("MSAMD_MATERIALISE_LOOKUPS")) continue;  // stage 2 reads the trace itself
    lk.mult = DBuf<u64>(ctx, h * c.num_lookups);
    lk.args = DBuf<u64>(ctx, std::max<size_t>(h * c.args_width, 1));
    if (!c.prefix_on_device || !lookup_values_device(ctx, c.prefix_prog, w->traces[ci].p, c.pre_width ? c.d_preprocessed.p : nullptr, h,
                                                     c.main_width, c.pre_width, c.args_width, lk.mult.p, lk.args.p))
      throw std::runtime_error("device-resident witness: this circuit's lookup prefix does not fit the device sweep");
    ctx.sync();
  }
  w->claim_offsets.resize(n_claims + 1);
  w->claim_data.resize(claim_elems);
  ctx.d2h(w->claim_offsets.data(), d_claim_offsets.p, (n_claims + 1) * 8);
  if (claim_elems) ctx.d2h(w->claim_data.data(), d_claim_data.p, claim_elems * 8);
  w->claim_elems_total = claim_elems;
  w->d_claim_offsets = std::move(d_claim_offsets);
  w->d_claim_data = std::move(d_claim_data);
  return w;
}

std::unique_ptr<HWitness> witness_create(HSystem& sys, const u64* const* traces, const u64* heights, const u64* const* mult,
                                         const u64* const* args, size_t n_claims, const u64* claim_offsets, const u64* claim_data) {
  Ctx& ctx = *sys.ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  std::unique_ptr<HWitness> w(new HWitness());
  w->sys = &sys;
  size_t C = sys.circuits.size();
  w->traces.resize(C);
  w->lookups.resize(C);
  for (size_t ci = 0; ci < C; ci++) {
    const HCircuit& c = sys.circuits[ci];
    size_t h = heights[ci];
    w->heights.push_back(h);
    if (h == 0) continue;
    if (h & (h - 1)) throw std::runtime_error("trace height must be a power of two");
    if (log2_strict(h) > NTT_MAX_LOG || log2_strict(h) + sys.params.log_blowup > TW_LOG)
      throw std::runtime_error("trace height exceeds the supported maximum");
    if (c.pre_width && h != c.pre_height) throw std::runtime_error("main trace height must equal preprocessed trace height");
    if (!traces[ci]) {  // computed by another rank (ms_prove_sharded): only the height is known here
      w->has_remote = true;
      continue;
    }
    size_t cnt = h * c.main_width;
    for (size_t i = 0; i < cnt; i++)
      if (traces[ci][i] >= GL_P) throw std::runtime_error("non-canonical trace value");
    w->traces[ci] = DBuf<u64>(ctx, cnt);
    ctx.h2d(w->traces[ci].p, traces[ci], cnt * 8);
    DLookups& lk = w->lookups[ci];
    lk.height = h;
    lk.num_lookups = c.num_lookups;
    lk.args_width = c.args_width;
    if (c.num_lookups == 0) continue;
    std::vector<uint32_t> offs(1, 0);
    for (auto& l : c.lookups) offs.push_back(offs.back() + (uint32_t)l.second.size());
    lk.arg_offsets = DBuf<uint32_t>(ctx, offs.size());
    ctx.h2d(lk.arg_offsets.p, offs.data(), offs.size() * 4);
    // from_stage_1 requested and the circuit has a fused stage-2 kernel: the lookup values are never materialised (the
    // kernel evaluates the lookup expressions from the trace; MSAMD_MATERIALISE_LOOKUPS=1 keeps the separate pass)
    if (!(mult && mult[ci]) && c.stage2_trace_jit.function && !getenv("MSAMD_MATERIALISE_LOOKUPS") && !getenv("MSAMD_HOST_LOOKUP_VALUES")) continue;
    lk.mult = DBuf<u64>(ctx, h * c.num_lookups);
    lk.args = DBuf<u64>(ctx, std::max<size_t>(h * c.args_width, 1));
    if (mult && mult[ci]) {
      ctx.h2d(lk.mult.p, mult[ci], h * c.num_lookups * 8);
      if (c.args_width) ctx.h2d(lk.args.p, args[ci], h * c.args_width * 8);
      ctx.sync();
    } else if (c.prefix_on_device && !getenv("MSAMD_HOST_LOOKUP_VALUES") &&
               lookup_values_device(ctx, c.prefix_prog, w->traces[ci].p, c.pre_width ? c.d_preprocessed.p : nullptr, h, c.main_width,
                                    c.pre_width, c.args_width, lk.mult.p, lk.args.p)) {
      ctx.sync();  // SystemWitness::from_stage_1 ran as one kernel
    } else {
      // SystemWitness::from_stage_1, src/system.rs:275-328 (host sweep: huge prefixes or malformed programs)
      std::vector<u64> hm, ha;
      host_lookup_values(c, traces[ci], h, hm, ha);
      ctx.h2d(lk.mult.p, hm.data(), hm.size() * 8);
      if (!ha.empty()) ctx.h2d(lk.args.p, ha.data(), ha.size() * 8);
      ctx.sync();
    }
  }
  w->claim_offsets.assign(claim_offsets, claim_offsets + n_claims + 1);
  size_t tot = n_claims ? (size_t)claim_offsets[n_claims] : 0;
  if (claim_offsets[0] != 0) throw std::runtime_error("claim offsets must start at 0");
  for (size_t i = 0; i < n_claims; i++)
    if (claim_offsets[i + 1] < claim_offsets[i]) throw std::runtime_error("claim offsets must be non-decreasing");
  w->claim_data.assign(claim_data, claim_data + tot);
  w->claim_elems_total = tot;
  for (u64 x : w->claim_data)
    if (x >= GL_P) throw std::runtime_error("non-canonical claim value");
  w->d_claim_offsets = DBuf<u64>(ctx, n_claims + 1);
  w->d_claim_data = DBuf<u64>(ctx, std::max<size_t>(tot, 1));
  ctx.h2d(w->d_claim_offsets.p, w->claim_offsets.data(), (n_claims + 1) * 8);
  if (tot) ctx.h2d(w->d_claim_data.p, w->claim_data.data(), tot * 8);
  ctx.sync();
  return w;
}

// ------------------------------------------------------------------ narrowing a host trace before its upload
// values[i] -> `pb`-byte little-endian words; returns the OR of everything read (the caller checks the bits above 8 pb);
// pack_host.cpp (plain host C++: AVX-512 where the CPU has it)
u64 narrow_range(const u64* in, uint8_t* out, unsigned pb, size_t n);
namespace {
// A handful of persistent host threads (MSAMD_PACK_THREADS, default 16; the process keeps them for its lifetime). One job at
// a time: the element range is cut into chunks, every chunk into one piece per worker; pieces are claimed in order, so
// the chunks complete in order and the caller uploads chunk k while the workers narrow chunk k + 1.
// the CPUs of the NUMA node that holds the page at `addr`, intersected with what this process may run on (empty: unknown)
std::vector<int> cpus_near(const void* addr) {
  std::vector<int> out;
  int node = -1;
#if defined(SYS_get_mempolicy)
  if (!addr || syscall(SYS_get_mempolicy, &node, nullptr, 0UL, const_cast<void*>(addr), 3UL /* MPOL_F_NODE | MPOL_F_ADDR */) != 0) node = -1;
#endif
  if (node < 0) return out;
  char path[96];
  snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
  FILE* f = fopen(path, "r");
  if (!f) return out;
  char buf[4096];
  const size_t got = fread(buf, 1, sizeof(buf) - 1, f);
  fclose(f);
  buf[got] = 0;
  cpu_set_t allowed;
  CPU_ZERO(&allowed);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return out;
  for (char* p = buf; *p;) {  // "0-63,128-191"
    char* e = nullptr;
    long a = strtol(p, &e, 10);
    if (e == p) break;
    long b = a;
    if (*e == '-') {
      p = e + 1;
      b = strtol(p, &e, 10);
    }
    for (long c = a; c <= b && c < CPU_SETSIZE; c++)
      if (CPU_ISSET((int)c, &allowed)) out.push_back((int)c);
    p = *e == ',' ? e + 1 : e;
    if (*e != ',') break;
  }
  return out;
}

class PackPool {
 public:
  // One pool per DEVICE (created on first use, never destroyed): a process that drives several GPUs from thread ranks
  // (ms_comm_local_*, `bench.py --gpus N` without a launcher) narrows every rank's trace at the same time instead of one job
  // after the other - a pool runs one job at a time. Contexts that share a device share its pool, as before.
  // `near`: an address inside the first trace that will be narrowed. With MSAMD_PACK_AFFINITY=1 the workers are kept on
  // the NUMA node that holds it; by default the scheduler places them
  static PackPool* get(int device, const void* near = nullptr) {
    static std::mutex mu;
    static std::map<int, PackPool*> pools;
    static const int n_threads = []() {
      int n = 16;
      if (const char* e = getenv("MSAMD_PACK_THREADS")) n = atoi(e);
      return n > 64 ? 64 : n;
    }();
    if (n_threads <= 0) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    auto it = pools.find(device);
    if (it != pools.end()) return it->second;
    // (measured: left to the scheduler the workers do as well or better - 6.99 against 7.34 ms per proof on one box of the
    // pool, a tie on another; confining sixteen busy threads to the node of the trace crowds the runtime's own threads there.
    // Polling workers between proofs instead of sleeping ones was worse still: 12-19 ms stalls every twenty proofs.)
    PackPool* p = new PackPool(n_threads, getenv("MSAMD_PACK_AFFINITY") ? cpus_near(near) : std::vector<int>());
    pools[device] = p;
    return p;
  }
  static constexpr size_t MAX_CHUNKS = 64, MAX_SUB = 128;
  size_t n_chunks = 0;
  // starts narrowing in[0 .. cnt) into out; chunk k covers elements [chunk_begin(k), chunk_begin(k + 1))
  // (run_words != 0, "row groups": `in` is a sequence of blocks of run_stride words and chunk k is made of the k-th run of
  // run_words words of every block, packed back to back: out[chunk_begin(k) + m * run_words + i] = in[m * run_stride + k * run_words + i].
  // run_stride = chunks * run_words, cnt a multiple of run_stride.)
  void start(const u64* in, uint8_t* out, unsigned pb, size_t cnt, size_t chunks, size_t run_words = 0) {
    job_mu_.lock();  // one job at a time (contexts on other threads wait here); released by finish()
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return busy_ == 0; });  // the previous job's last workers have left (finish() does not wait for them)
    in_ = in, out_ = out, pb_ = pb, cnt_ = cnt;
    caller_ = std::this_thread::get_id();
    straggle_us_ = getenv("MSAMD_PACK_STRAGGLE_US") ? (unsigned)atoi(getenv("MSAMD_PACK_STRAGGLE_US")) : 0;  // (per job: tests flip it)
    wait_all_ = getenv("MSAMD_PACK_WAIT_ALL") != nullptr;
    n_chunks = std::max<size_t>(1, std::min(chunks, MAX_CHUNKS));
    run_words_ = run_words;
    if (run_words) {
      per_chunk_ = cnt / n_chunks;
      run_stride_ = n_chunks * run_words;
      runs_per_chunk_ = cnt / run_stride_;
    } else {
      per_chunk_ = ((cnt + n_chunks - 1) / n_chunks + 63) & ~size_t(63);
      n_chunks = (cnt + per_chunk_ - 1) / per_chunk_;
    }
    next_.store(0);
    acc_.store(0);
    nsub_ = std::min<size_t>(2 * threads_.size(), MAX_SUB);  // pieces per chunk
    for (size_t k = 0; k < n_chunks; k++) left_[k].store((int)nsub_);
    for (size_t i = 0; i < n_chunks * nsub_; i++) state_[i].store(0, std::memory_order_relaxed);
    busy_ = (int)threads_.size();
    gen_++;
    cv_.notify_all();
  }
  size_t chunk_begin(size_t k) const { return std::min(cnt_, k * per_chunk_); }
  // waits until chunk k is complete; false = some value read so far does not fit `pb` bytes
  // (the caller narrows pieces itself while it waits: it is awake, the workers may still be waking up. A piece whose worker
  // has been descheduled - the host cores are shared with the pool's other tenants - does not hold the chunk up either: once
  // every piece of the chunk has been claimed, pieces still in work after a grace period are narrowed AGAIN by the caller; both
  // write the same bytes, whoever finishes first counts)
  bool wait_chunk(size_t k) {
    double all_claimed_at = 0;
    while (left_[k].load(std::memory_order_acquire) != 0) {
      if (next_.load(std::memory_order_relaxed) >= (k + 1) * nsub_) {  // every piece of this chunk has a taker
        const double now = now_ms();
        if (all_claimed_at == 0) all_claimed_at = now;
        if (now - all_claimed_at > 0.03 && !wait_all_) {  // 30 us: a healthy worker finishes a piece in 3-5 us
          for (size_t sub = 0; sub < nsub_ && left_[k].load(std::memory_order_acquire) != 0; sub++)
            if (state_[k * nsub_ + sub].load(std::memory_order_acquire) != 2) work_piece(k * nsub_ + sub, true);  // (claimed: in work, or its worker not even started)
          continue;
        }
      }
      if (!take_piece()) std::this_thread::yield();
    }
    return (acc_.load() >> (8 * pb_)) == 0;
  }
  // the job is over for its caller. Workers that are late (still waking up, or descheduled inside a piece the caller has redone)
  // leave on their own: the next start() and quiesce() wait for them (so a SECOND large trace of the same proof still waits for
  // the first one's late worker: one job at a time)
  void finish() {
    if (wait_all_) quiesce();  // MSAMD_PACK_WAIT_ALL=1: as before round 4's last change (every worker checks out, no second taker)
    job_mu_.unlock();
  }
  // no worker is inside a job (the buffers of finished jobs may be freed)
  void quiesce() {
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [&] { return busy_ == 0; });
  }
  struct Job {  // start ... finish, also when an error unwinds the caller
    PackPool& p;
    Job(PackPool& pool, const u64* in, uint8_t* out, unsigned pb, size_t cnt, size_t chunks, size_t run_words = 0) : p(pool) {
      p.start(in, out, pb, cnt, chunks, run_words);
    }
    ~Job() { p.finish(); }
  };

 private:
  PackPool(int n, const std::vector<int>& cpus) {
    for (int i = 0; i < n; i++) threads_.emplace_back([this]() { run(); });
    if ((int)cpus.size() >= n) {
      cpu_set_t set;
      CPU_ZERO(&set);
      for (int c : cpus) CPU_SET(c, &set);
      for (auto& t : threads_) (void)pthread_setaffinity_np(t.native_handle(), sizeof(set), &set);  // best effort
    }
    for (auto& t : threads_) t.detach();
  }
  // claims the next piece of the job and narrows it; false = every piece has been claimed
  bool take_piece() {
    const size_t it = next_.fetch_add(1);
    if (it >= n_chunks * nsub_) return false;
    work_piece(it, false);
    return true;
  }
  // narrows piece `it` (again: the caller's second go at a piece whose worker is late). The piece counts once.
  void work_piece(size_t it, bool again) {
    const size_t k = it / nsub_, sub = it % nsub_;
    if (!again) {
      uint8_t unclaimed = 0;
      if (!state_[it].compare_exchange_strong(unclaimed, 1, std::memory_order_acq_rel)) return;  // the caller has taken this piece over already
      if (straggle_us_ && sub == 3 && std::this_thread::get_id() != caller_)  // diagnostics: one piece per chunk is held back on its worker
        std::this_thread::sleep_for(std::chrono::microseconds(straggle_us_));
    }
    u64 acc = 0;
    if (run_words_) {
      const size_t per = (runs_per_chunk_ + nsub_ - 1) / nsub_;
      const size_t m0 = std::min(runs_per_chunk_, sub * per), m1 = std::min(runs_per_chunk_, m0 + per);
      for (size_t m = m0; m < m1; m++)
        acc |= narrow_range(in_ + m * run_stride_ + k * run_words_, out_ + (k * per_chunk_ + m * run_words_) * pb_, pb_, run_words_);
    } else {
      const size_t b = chunk_begin(k), e = chunk_begin(k + 1);
      const size_t piece = (((e - b) + nsub_ - 1) / nsub_ + 7) & ~size_t(7);
      const size_t lo = std::min(e, b + sub * piece), hi = std::min(e, lo + piece);
      if (hi > lo) acc = narrow_range(in_ + lo, out_ + lo * pb_, pb_, hi - lo);
    }
    if (acc) acc_.fetch_or(acc);
    if (state_[it].exchange(2, std::memory_order_acq_rel) != 2) left_[k].fetch_sub(1, std::memory_order_release);
  }
  void run() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
      }
      while (take_piece()) {
      }
      {
        std::unique_lock<std::mutex> lk(mu_);
        if (--busy_ == 0) done_cv_.notify_all();
      }
    }
  }
  std::vector<std::thread> threads_;
  std::mutex mu_, job_mu_;
  std::condition_variable cv_, done_cv_;
  uint64_t gen_ = 0;
  int busy_ = 0;
  const u64* in_ = nullptr;
  uint8_t* out_ = nullptr;
  unsigned pb_ = 1;
  size_t cnt_ = 0, per_chunk_ = 0, nsub_ = 1;
  size_t run_words_ = 0, run_stride_ = 0, runs_per_chunk_ = 0;
  std::atomic<uint8_t> state_[MAX_CHUNKS * MAX_SUB];  // per piece: 0 unclaimed, 1 in work, 2 done
  std::thread::id caller_;
  unsigned straggle_us_ = 0;  // MSAMD_PACK_STRAGGLE_US (diagnostics)
  bool wait_all_ = false;
  std::atomic<size_t> next_{0};
  std::atomic<u64> acc_{0};
  std::atomic<int> left_[MAX_CHUNKS];
};
}  // namespace

// ------------------------------------------------------------------ host-resident SystemWitness
// The reference's prove() is handed a witness that lives in host memory (benches/multi_stark.rs:292-296,
// src/prover.rs:290-295). Nothing is uploaded here: the caller's buffers are page-locked so that the per-proof uploads
// run at the link rate, the values are validated, and prove() moves them to HBM on the copy stream every time.
// page-locking of caller memory: counted per range process-wide (msamd.h host_range_pin)
void HWitness::pin(const void* p, size_t bytes) {
  if (!p || !bytes) return;
  const int r = host_range_pin(p, bytes);
  if (r == 1)
    registered.push_back(const_cast<void*>(p));
  else if (r == 0)
    pinned = false;  // uploads of this witness go through the context's bounce buffer
}
HWitness::~HWitness() {
  if (!host_resident) return;
  if (sys && sys->ctx) {
    (void)hipSetDevice(sys->ctx->device);
    (void)hipStreamSynchronize(sys->ctx->copy_stream);  // a prefetch may still be writing into the staged buffers
    (void)hipStreamSynchronize(sys->ctx->claims_stream);
    // (and nothing of an abandoned or delayed proof may still be queued anywhere when the caller's ranges lose their page lock)
    if (sys->ctx->side_stream) (void)hipStreamSynchronize(sys->ctx->side_stream);
    (void)hipStreamSynchronize(sys->ctx->main_stream);
  }
  for (auto& st : stage)
    for (auto& e : st.ev)
      if (e) (void)hipEventDestroy(e);
  bool any_packed = false;
  for (uint8_t* p : h_packed) any_packed = any_packed || p != nullptr;
  if (any_packed)
    if (PackPool* pool = PackPool::get(sys && sys->ctx ? sys->ctx->device : 0)) pool->quiesce();  // a late worker of the last proof may still be inside its piece
  for (void* p : registered) host_range_unpin(p);
  for (uint8_t* p : h_packed)
    if (p) (void)hipHostFree(p);
  (void)hipGetLastError();  // (a range the caller has already freed or re-registered: nothing to report, nothing to leave behind)
}

std::unique_ptr<HWitness> witness_create_host(HSystem& sys, const u64* const* traces, const u64* heights, size_t n_claims,
                                              const u64* claim_offsets, const u64* claim_data, size_t data_first, size_t data_count,
                                              const u64* head, size_t n_head) {
  Ctx& ctx = *sys.ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  std::unique_ptr<HWitness> w(new HWitness());
  w->sys = &sys;
  w->host_resident = true;
  const size_t C = sys.circuits.size();
  w->traces.resize(C);
  w->lookups.resize(C);
  w->h_traces.assign(C, nullptr);
  w->h_mult.resize(C);
  w->h_args.resize(C);
  w->pack_bytes.assign(C, 0);
  w->h_packed.assign(C, nullptr);
  size_t pack_min = size_t(4) << 20;  // below this the plain upload takes less than waking the threads
  if (const char* e = getenv("MSAMD_PACK_MIN_BYTES")) pack_min = (size_t)atoll(e);
  const void* first_trace = nullptr;  // the tallest one
  for (size_t ci = 0, best = 0; ci < C; ci++)
    if (heights[ci] > best && traces[ci]) first_trace = traces[ci], best = heights[ci];
  const bool may_pack = !getenv("MSAMD_NO_PACK") && PackPool::get(ctx.device, first_trace) != nullptr;
  for (size_t ci = 0; ci < C; ci++) {
    const HCircuit& c = sys.circuits[ci];
    const size_t h = heights[ci];
    w->heights.push_back(h);
    if (h == 0) continue;
    if (h & (h - 1)) throw std::runtime_error("trace height must be a power of two");
    if (log2_strict(h) > NTT_MAX_LOG || log2_strict(h) + sys.params.log_blowup > TW_LOG)
      throw std::runtime_error("trace height exceeds the supported maximum");
    if (c.pre_width && h != c.pre_height) throw std::runtime_error("main trace height must equal preprocessed trace height");
    if (!traces[ci]) {  // computed by another rank (ms_prove_sharded): only the height is known here
      w->has_remote = true;
      continue;
    }
    const size_t cnt = h * c.main_width;
    u64 seen = 0;
    for (size_t i = 0; i < cnt; i++) {
      if (traces[ci][i] >= GL_P) throw std::runtime_error("non-canonical trace value");
      seen |= traces[ci][i];
    }
    w->h_traces[ci] = traces[ci];
    w->pin(traces[ci], cnt * 8);
    if (may_pack && cnt * 8 >= pack_min && (seen >> 32) == 0) {
      const unsigned pb = (seen >> 8) == 0 ? 1 : (seen >> 16) == 0 ? 2 : 4;
      void* hp = nullptr;
      if (hipHostMalloc(&hp, cnt * pb, hipHostMallocDefault) == hipSuccess) {
        w->pack_bytes[ci] = pb;
        w->h_packed[ci] = (uint8_t*)hp;
      } else {
        (void)hipGetLastError();
      }
    }
    DLookups& lk = w->lookups[ci];
    lk.height = h;
    lk.num_lookups = c.num_lookups;
    lk.args_width = c.args_width;
    if (c.num_lookups == 0) continue;
    std::vector<uint32_t> offs(1, 0);
    for (auto& l : c.lookups) offs.push_back(offs.back() + (uint32_t)l.second.size());
    lk.arg_offsets = DBuf<uint32_t>(ctx, offs.size());
    ctx.h2d(lk.arg_offsets.p, offs.data(), offs.size() * 4);
    unsigned threads = 64;
    const bool fits = c.prefix_on_device && c.prefix_prog.n_slots * threads * 8 <= 64 * 1024 && !getenv("MSAMD_HOST_LOOKUP_VALUES");
    if (!fits) {  // the lookup values of this circuit are part of the host-resident witness (and of every upload)
      host_lookup_values(c, traces[ci], h, w->h_mult[ci], w->h_args[ci]);
      w->pin(w->h_mult[ci].data(), w->h_mult[ci].size() * 8);
      w->pin(w->h_args[ci].data(), w->h_args[ci].size() * 8);
    }
  }
  if (claim_offsets[0] != 0) throw std::runtime_error("claim offsets must start at 0");
  for (size_t i = 0; i < n_claims; i++)
    if (claim_offsets[i + 1] < claim_offsets[i]) throw std::runtime_error("claim offsets must be non-decreasing");
  const size_t tot = n_claims ? (size_t)claim_offsets[n_claims] : 0;
  w->claim_offsets.assign(claim_offsets, claim_offsets + n_claims + 1);
  w->claim_elems_total = tot;
  if (data_count == ~size_t(0)) {
    w->claim_data.assign(claim_data, claim_data + tot);
  } else {  // this rank's part of the data only
    if (data_first > tot || data_count > tot - data_first) throw std::runtime_error("claims slice outside the claims' data");
    const size_t need_head = std::min<size_t>(tot, 130);
    if (n_head < need_head || (need_head && !head)) throw std::runtime_error("claims slice: the first 130 elements are needed on every rank");
    w->claims_partial = true;
    w->has_remote = true;  // (only ms_prove_sharded accepts such a witness)
    w->claim_elem0 = data_first;
    w->claim_data.assign(claim_data, claim_data + data_count);
    w->claim_head.assign(head, head + need_head);
    for (u64 x : w->claim_head)
      if (x >= GL_P) throw std::runtime_error("non-canonical claim value");
  }
  for (u64 x : w->claim_data)
    if (x >= GL_P) throw std::runtime_error("non-canonical claim value");
  // the uploads read the witness's own copies (the small-claims transcript needs them on the host anyway)
  w->pin(w->claim_offsets.data(), (n_claims + 1) * 8);
  w->pin(w->claim_data.data(), w->claim_data.size() * 8);
  ctx.sync();
  return w;
}

namespace {
// Per-proof upload of a host-resident witness. Everything is queued on the copy stream in the order the proof needs it
// (traces, then claims); the kernels of ctx.stream wait on events, so the claims travel while stage 1 is computed.
// The device buffers live in the witness for the duration of the proof only. With prefetching on, the upload for the
// FOLLOWING proof is queued right behind this one's, into a second set of buffers, and travels while this proof is computed.
struct HostUpload {
  HWitness& w;
  Ctx& ctx;
  bool on = false;
  bool skip_claims = false;  // the multi-rank prover uploads per-rank slices of the claims itself
  // Row groups (plain prover only, opt-in: MSAMD_ROW_GROUPS=1). A transform needs every row of a column, so with chunks of
  // consecutive rows nothing of stage 1 can start before the last chunk has landed. The first pass of the inverse transform,
  // though, works on 4096-row tiles of the bit-reversed storage, and tile t holds the natural rows r with r mod 2^T = rev_T(t),
  // T = log2 h - 12: a chunk made of the rows whose residue has k in its top three bits - runs of 2^(T-3) consecutive rows
  // every 2^T - completes the tiles t = 8 i + rev3(k). So chunk k is narrowed, pulled (scattered back to its rows: stage 2 reads
  // the row-major trace), transposed and taken through that first pass while the host narrows chunk k + 1, and stage 1 starts
  // at the second pass: 0.12 ms of kernels leave the critical path. Measured (2^20 x 14, 32-row runs of 3.5 KB every 28 KB):
  // the HOST side loses more than that - sixteen threads narrow the trace in 0.39 ms from consecutive rows and in 0.52-0.59 ms
  // from such runs (tools/micro/pack_runs.cpp; 0.43 ms only at 512-row runs, which would need a first pass of 8 levels), the
  // last chunk is queued at 0.94-1.00 ms instead of 0.41-0.48, and the step is 0.05-0.2 ms SLOWER. Hence off by default.
  bool row_groups = false;
  HostUpload(HWitness& wit, Ctx& c) : w(wit), ctx(c) {}
  // does stage 2 of this circuit run from the uploaded trace (stage2_terms_trace_jit)?
  bool fused(size_t ci) const {
    const HCircuit& c = w.sys->circuits[ci];
    return w.host_resident && c.stage2_trace_jit.function && w.h_mult[ci].empty();
  }
  void issue(HWitness::Staged& st) {
    HSystem& sys = *w.sys;
    const size_t C = sys.circuits.size();
    for (auto& e : st.ev)
      if (!e) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    st.clear();
    st.traces.resize(C);
    st.evals.resize(C);
    st.mult.resize(C);
    st.args.resize(C);
    // buffers first: pool blocks handed out here may still be in use by kernels queued earlier on ctx.stream
    HIP_CHECK(hipEventRecord(ctx.copy_ev[3], ctx.stream));
    HIP_CHECK(hipStreamWaitEvent(ctx.copy_stream, ctx.copy_ev[3], 0));
    ctx.side_config();
    ctx.copy_delay();  // (diagnostics: MSAMD_COPY_DELAY_US)
    std::vector<DBuf<uint8_t>> narrow(C);  // released when the proof's uploads have been waited for (the destructor's sync)
    bool used_second = false;
    HIP_CHECK(hipStreamWaitEvent(ctx.claims_stream, ctx.copy_ev[3], 0));  // (pool blocks may still be in use by earlier kernels)
    // The claims (42 MB at the bench size) are needed when the stage-1 tree is hashed, 0.8 ms after the trace has landed, and
    // take 0.75 ms of the link: behind the trace's chunks on the copy stream they arrive just in time (kernel-only rocprofv3
    // timeline: claims_words_k starts with the leaf hashing). MSAMD_CLAIMS_OWN_STREAM=1 starts them at once on a stream of
    // their own, as DMA copies beside the chunks' pulling kernels - measured WORSE (6.9 against 6.4 ms per step): the link is the
    // narrow upload's bottleneck, and what the claims take of it early delays the trace, which is on the critical path.
    const size_t n_claims = w.claim_offsets.size() - 1, tot = w.claim_data.size();
    // Uploads read the caller's page-locked ranges (HWitness::pin). Where a range could not be locked, the words go through the
    // context's bounce buffer instead of an asynchronous copy from pageable memory (Ctx::bounce_h2d has the reason)
    auto from_caller = [&](void* dst, const void* src, size_t bytes, hipStream_t s) {
      if (w.pinned)
        HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
      else
        ctx.bounce_h2d(dst, src, bytes, s);
    };
    static const bool claims_own_stream = getenv("MSAMD_CLAIMS_OWN_STREAM") != nullptr;
    auto upload_claims = [&](hipStream_t s) {
      st.claim_offsets = DBuf<u64>(ctx, n_claims + 1);
      st.claim_data = DBuf<u64>(ctx, std::max<size_t>(tot, 1));
      from_caller(st.claim_offsets.p, w.claim_offsets.data(), (n_claims + 1) * 8, s);
      if (tot) from_caller(st.claim_data.p, w.claim_data.data(), tot * 8, s);
      HIP_CHECK(hipEventRecord(st.ev[2], s));
    };
    if (!skip_claims && claims_own_stream) {
      HIP_CHECK(hipStreamWaitEvent(ctx.claims_stream, ctx.copy_ev[3], 0));  // (the pool blocks may still be in use, as above)
      upload_claims(ctx.claims_stream);
    }
    for (size_t ci = 0; ci < C; ci++) {
      const HCircuit& c = sys.circuits[ci];
      const size_t h = w.heights[ci];
      if (!h || !w.h_traces[ci]) continue;  // inactive, or computed by another rank
      const size_t cnt = h * c.main_width;
      st.traces[ci] = DBuf<u64>(ctx, cnt);
      const unsigned pb = w.pack_bytes[ci];
      PackPool* pool = pb && !w.prefetch ? PackPool::get(ctx.device) : nullptr;  // (a prefetch already travels behind the running proof)
      bool sent = false;
      if (pool) {
        // the host threads narrow chunk k + 1 while chunk k crosses the link; the device widens the whole trace afterwards
        static const bool pull = !getenv("MSAMD_NO_PULL");  // MSAMD_NO_PULL=1: DMA copy into a staging buffer + widening launch
        static const bool two_streams = getenv("MSAMD_PULL_TWO_STREAMS") != nullptr;  // (measured: no difference, 6.03-6.16 ms either way)
        if (!pull) narrow[ci] = DBuf<uint8_t>(ctx, cnt * pb);
        const bool want_groups = getenv("MSAMD_ROW_GROUPS") != nullptr;  // (read per proof: tests flip it)
        const unsigned logh = log2_strict(h);
        const bool groups = row_groups && pull && !two_streams && want_groups && (size_t(1) << logh) == h && logh >= 19 && cnt < (size_t(1) << 35) &&
                            ctx.stream == ctx.main_stream;
        if (groups) {
          const size_t T = logh - 12, run_rows = size_t(1) << (T - 3), run_words = run_rows * c.main_width, run_stride = run_words * 8;
          st.evals[ci] = DBuf<u64>(ctx, cnt);
          PackPool::Job job(*pool, w.h_traces[ci], w.h_packed[ci], pb, cnt, 8, run_words);
          sent = true;
          auto first_pass = [&](unsigned k) {  // group k has landed: transposed and through the first pass on the main stream
            HIP_CHECK(hipStreamWaitEvent(ctx.stream, ctx.group_event(k), 0));
            transpose_in_rows_part(ctx, st.traces[ci].p, st.evals[ci].p, h, c.main_width, k);
            ntt_dit_first_pass_part(ctx, st.evals[ci].p, logh, c.main_width, true, bitrev32(k, 3));
          };
          for (unsigned k = 0; k < 8; k++) {
            if (!pool->wait_chunk(k)) {  // a value outgrew the width found at creation: the plain path below
              sent = false;
              break;
            }
            if (k == 0) g_probes.mark("narrow upload: first chunk ready");
            pull_widen_runs(w.h_packed[ci] + pool->chunk_begin(k) * pb, pb, cnt / 8, st.traces[ci].p + k * run_words, run_words, run_stride, ctx.copy_stream);
            HIP_CHECK(hipEventRecord(ctx.group_event(k), ctx.copy_stream));
            if (k) first_pass(k - 1);  // (behind the pull: the link must not wait for this thread's launches)
          }
          if (sent) first_pass(7);
          g_probes.mark("narrow upload: last chunk queued");
          if (!sent) st.evals[ci].reset();  // (what the launches above wrote is abandoned; the block is reused behind them)
        } else {
          static const size_t n_chunks = getenv("MSAMD_PACK_CHUNKS") ? (size_t)atoi(getenv("MSAMD_PACK_CHUNKS")) : 8;
          PackPool::Job job(*pool, w.h_traces[ci], w.h_packed[ci], pb, cnt, n_chunks);
          sent = true;
          for (size_t k = 0; k < pool->n_chunks; k++) {
            if (!pool->wait_chunk(k)) {  // a value outgrew the width found at creation: the plain path below
              sent = false;
              break;
            }
            if (k == 0) g_probes.mark("narrow upload: first chunk ready");
            const size_t b = pool->chunk_begin(k), e = pool->chunk_begin(k + 1);
            if (pull) {  // ONE launch per chunk: the kernel reads the pinned narrow words over the link and writes 64-bit words
              // (two streams in turn: the next chunk's first requests leave while this chunk's last ones drain)
              const bool second = two_streams && (k & 1);
              pull_widen_words(w.h_packed[ci] + b * pb, pb, e - b, st.traces[ci].p + b, second ? ctx.claims_stream : ctx.copy_stream);
              used_second = used_second || second;
            } else {
              HIP_CHECK(hipMemcpyAsync(narrow[ci].p + b * pb, w.h_packed[ci] + b * pb, (e - b) * pb, hipMemcpyHostToDevice, ctx.copy_stream));
              widen_words(narrow[ci].p + b * pb, pb, e - b, st.traces[ci].p + b, ctx.copy_stream);  // behind its chunk: only the last one is exposed
            }
          }
          g_probes.mark("narrow upload: last chunk queued");
        }
      }
      if (!sent) from_caller(st.traces[ci].p, w.h_traces[ci], cnt * 8, ctx.copy_stream);
    }
    if (used_second) {  // the chunks pulled on the second stream are part of "the traces have arrived"
      HIP_CHECK(hipEventRecord(ctx.copy_ev[2], ctx.claims_stream));
      HIP_CHECK(hipStreamWaitEvent(ctx.copy_stream, ctx.copy_ev[2], 0));
    }
    HIP_CHECK(hipEventRecord(st.ev[0], ctx.copy_stream));
    st.narrow = std::move(narrow);
    st.has_host_lookups = false;
    for (size_t ci = 0; ci < C; ci++) {
      const HCircuit& c = sys.circuits[ci];
      const size_t h = w.heights[ci];
      if (!h || !c.num_lookups || !w.h_traces[ci]) continue;
      if (fused(ci)) continue;  // stage 2 reads the trace itself: no LookupValues for this circuit
      st.mult[ci] = DBuf<u64>(ctx, h * c.num_lookups);
      st.args[ci] = DBuf<u64>(ctx, std::max<size_t>(h * c.args_width, 1));
      if (!w.h_mult[ci].empty()) {
        st.has_host_lookups = true;
        from_caller(st.mult[ci].p, w.h_mult[ci].data(), w.h_mult[ci].size() * 8, ctx.copy_stream);
        if (!w.h_args[ci].empty()) from_caller(st.args[ci].p, w.h_args[ci].data(), w.h_args[ci].size() * 8, ctx.copy_stream);
      }
    }
    if (!skip_claims && !claims_own_stream) upload_claims(ctx.copy_stream);
    if (skip_claims) HIP_CHECK(hipEventRecord(st.ev[2], ctx.copy_stream));
    // SystemWitness::from_stage_1 (src/system.rs:244-328) as one kernel per circuit, from the traces just uploaded, queued on
    // the copy stream behind the copies: it writes 344 MB at config 2 and runs beside the transforms of stage 1 (which are
    // bound by the vector ALU) instead of in front of stage 2
    for (size_t ci = 0; ci < C; ci++) {
      const HCircuit& c = sys.circuits[ci];
      const size_t h = w.heights[ci];
      if (!h || !c.num_lookups || !w.h_mult[ci].empty() || fused(ci) || !w.h_traces[ci]) continue;
      if (!lookup_values_device(ctx, c.prefix_prog, st.traces[ci].p, c.pre_width ? c.d_preprocessed.p : nullptr, h, c.main_width, c.pre_width,
                                c.args_width, st.mult[ci].p, st.args[ci].p, ctx.copy_stream))
        throw std::runtime_error("host-resident witness: lookup prefix does not fit the device sweep");
    }
    HIP_CHECK(hipEventRecord(st.ev[1], ctx.copy_stream));
    st.valid = true;
  }
  void start() {
    if (!w.host_resident) return;
    on = true;
    if (w.prefetch && w.stage[w.cur ^ 1].valid)
      w.cur ^= 1;  // the previous proof has already brought this one's inputs over
    else
      issue(w.stage[w.cur]);
    HWitness::Staged& st = w.stage[w.cur];
    // the proof reads through the witness's usual members
    w.early_evals.resize(st.traces.size());
    for (size_t ci = 0; ci < st.traces.size(); ci++) {
      w.traces[ci] = std::move(st.traces[ci]);
      w.early_evals[ci] = std::move(st.evals[ci]);
      w.lookups[ci].mult = std::move(st.mult[ci]);
      w.lookups[ci].args = std::move(st.args[ci]);
    }
    w.d_claim_offsets = std::move(st.claim_offsets);
    w.d_claim_data = std::move(st.claim_data);
    st.valid = false;
    if (w.prefetch) issue(w.stage[w.cur ^ 1]);
  }
  void wait_traces() {
    if (on) HIP_CHECK(hipStreamWaitEvent(ctx.stream, w.stage[w.cur].ev[0], 0));
  }
  // the lookup values (uploaded, or computed on the copy stream from the uploaded traces) are complete
  void lookup_values() {
    if (!on) return;
    HIP_CHECK(hipStreamWaitEvent(ctx.stream, w.stage[w.cur].ev[1], 0));
    // a side stream that is already forked (the claims' logUp sum runs there) does not inherit this wait: the short circuits'
    // stage-2 kernels queued on it next read the same lookup values
    if (ctx.side_forked && ctx.stream != ctx.side_stream) HIP_CHECK(hipStreamWaitEvent(ctx.side_stream, w.stage[w.cur].ev[1], 0));
  }
  void wait_claims() {
    if (on) HIP_CHECK(hipStreamWaitEvent(ctx.stream, w.stage[w.cur].ev[2], 0));
  }
  ~HostUpload() {
    if (!on) return;
    // this proof's copies may still be in flight when it is abandoned: wait before the blocks return to the pool (a
    // prefetch queued behind them is then complete as well, which costs nothing: it is shorter than the proof)
    (void)hipStreamSynchronize(ctx.copy_stream);
    (void)hipStreamSynchronize(ctx.claims_stream);
    w.stage[w.cur].narrow.clear();
    for (auto& t : w.traces) t.reset();
    for (auto& t : w.early_evals) t.reset();
    for (auto& lk : w.lookups) {
      lk.mult.reset();
      lk.args.reset();
    }
    w.d_claim_offsets.reset();
    w.d_claim_data.reset();
  }
};
}  // namespace

// ------------------------------------------------------------------ proof bytes (Proof::to_bytes, src/prover.rs:241-248)
namespace {
struct PW {
  std::vector<uint8_t> b;
  void u8(uint8_t x) { b.push_back(x); }
  void u64_(u64 x) {
    size_t o = b.size();
    b.resize(o + 8);
    memcpy(&b[o], &x, 8);
  }
  void ext(E2 e) {
    u64_(e.c0);
    u64_(e.c1);
  }
  void raw(const void* p, size_t n) {
    const uint8_t* q = (const uint8_t*)p;
    b.insert(b.end(), q, q + n);
  }
  void cap(const std::vector<Digest>& c) {
    u64_(c.size());
    for (auto& d : c) raw(d.b, 32);
  }
};
typedef std::vector<std::vector<std::vector<E2>>> OpenedRound;
void write_round(PW& w, const OpenedRound& r) {
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
  PcsData* data;
  std::vector<std::vector<E2>> points;
};

bool e2_same(E2 a, E2 b) { return a.c0 == b.c0 && a.c1 == b.c1; }

// Opening points that only the DEVICE knows while the opening's first kernels are queued (the device transcript sampled
// zeta, outer.hip): a point is named by a placeholder (sym_point(id): c1 is not a canonical field element, so it never equals
// a real point), its value lies at d_points[id], and `resolve` - called right after the opened-value read-back, the first
// moment the host needs the values - returns every value; pcs_open then swaps the placeholders for them.
struct SymbolicPoints {
  const E2* d_points = nullptr;
  const u32* d_state = nullptr;  // 8 words: the challenger's input buffer behind the last point's sample (device), or null
  size_t n = 0;
  std::function<void(std::vector<E2>& values)> resolve;
  std::vector<int> next_log;  // per id: point id = point 0 times the generator of the subgroup of order 2^next_log[id] (-1: unrelated)
};
inline E2 sym_point(size_t id) { return e2((u64)id, ~u64(0)); }
inline bool is_sym_point(E2 z) { return z.c1 == ~u64(0); }

// MSAMD_TRACE=1: synchronise and print the wall time of each phase of the opening (diagnostics only)
struct PhaseTrace {
  Ctx& ctx;
  bool on;
  double t;
  explicit PhaseTrace(Ctx& c) : ctx(c), on(getenv("MSAMD_TRACE") != nullptr), t(0) {
    if (on) {
      ctx.sync();
      t = now_ms();
    }
  }
  void mark(const char* what) {
    if (!on) return;
    ctx.sync();
    double n = now_ms();
    fprintf(stderr, "[msamd] %-22s %8.3f ms\n", what, n - t);
    t = n;
  }
};

// challenger.grind(bits): the search runs on the device when the pending transcript is a single BLAKE3 chunk
u64 grind(Ctx& ctx, Challenger& ch, unsigned bits) {
  if (bits == 0) return 0;
  u64 w = 0;
  if (bits >= 4 && grind_device(ctx, ch.input, bits, &w)) {
    ch.observe(w);
    if (ch.sample_bits(bits) != 0) throw std::runtime_error("grind: device witness rejected by the host challenger");
    return w;
  }
  return ch.grind(bits);
}

// shape of the input rounds' openings inside one query: per round the matrix widths and the number of siblings
struct InputShape {
  std::vector<std::vector<size_t>> widths;
  std::vector<size_t> nsib;
};
struct InputGather {
  // optional: called with the query indices in DEVICE memory as soon as the device has sampled them, before the FRI phase's one
  // synchronisation - what it queues (gathers, exchanges, read-backs) arrives with that synchronisation
  std::function<void(const u64* d_indices, size_t nq)> queue;
  // the openings of the input rounds for the (host-checked) indices, one block of qbytes per query
  std::function<void(const std::vector<uint64_t>& indices, size_t qbytes, uint8_t* out)> fetch;
};
void add_gather_seg(std::vector<GatherSeg>& segs, size_t& out_off, const void* base, u64 stride, uint32_t count, uint32_t kind,
                    uint32_t shift, uint32_t flip) {
  GatherSeg q;
  q.base = base;
  q.stride = stride;
  q.count = count;
  q.kind = kind;
  q.shift = shift;
  q.flip = flip;
  q.out_off = out_off;
  segs.push_back(q);
  out_off += kind == 0 || kind == 3 ? size_t(count) * 8 : kind == 1 ? 32 : 16;
}
// Commit-phase rounds that ran BEFORE fri_prove on row shards spread over ranks (prover_sharded.inc): their commitments
// and proof-of-work witnesses go in front of fri_prove's own, the query indices keep the full height, and each query's
// openings of those rounds (sibling value, then the path) arrive through `remote`, behind the input rounds' bytes.
struct FriHead {
  unsigned n_rounds = 0;
  unsigned log_max_height = 0;                 // log2 of the tallest reduced-opening vector (what the indices are sampled for)
  std::vector<std::vector<Digest>> commits;    // per head round
  std::vector<u64> pow;
  std::vector<size_t> nsib;                    // path length of each head round's opening
  // The head rounds' transcript steps ran on the DEVICE (merkle_top_challenge): d_state is the challenger state they left
  // (fri_prove goes on from it), d_recs their records (root, witness, beta); fri_prove reads the records back with its own,
  // replays them on the host challenger and fills `commits` and `pow` - which are empty until then.
  bool on_device = false;
  DBuf<uint32_t> d_state;
  DBuf<FriTailRound> d_recs;
  size_t qbytes() const {
    size_t b = 0;
    for (size_t n : nsib) b += 16 + 32 * n;
    return b;
  }
};
// FRI's transcript starts from a state that lies on the DEVICE (the opened values were absorbed there, open_alpha_k): fri_prove
// takes it from d_state instead of uploading the host challenger's, and calls after_wait right behind its first wait - the
// host catches up there (replays what the device did, checks it) before FRI's own steps are replayed.
struct FriDeviceStart {
  DBuf<uint32_t> d_state;
  std::function<void()> after_wait;
};
void fri_prove(HSystem& sys, Challenger& ch, std::vector<DBuf<E2>>& inputs, unsigned log_gmax, const std::vector<GatherSeg>& input_segs,
               size_t input_qbytes, const InputShape& shape, const InputGather* remote, PW& fri_bytes, PhaseTrace& tr,
               FriHead* head = nullptr, DTree* round0 = nullptr, FriDeviceStart* dstart = nullptr);

// TwoAdicFriPcs::open + prove_fri; serialises the FriProof straight into `fri_bytes`.
void pcs_open(HSystem& sys, std::vector<OpenRound>& rounds, Challenger& ch, std::vector<OpenedRound>& opened, PW& fri_bytes,
              const SymbolicPoints* sym = nullptr) {
  Ctx& ctx = *sys.ctx;
  const Params& prm = sys.params;
  const unsigned lb = (unsigned)prm.log_blowup;
  size_t gmax = 0, gw = 0;
  for (auto& r : rounds)
    for (auto& m : r.data->ldes) {
      gmax = std::max(gmax, m.h);
      gw = std::max(gw, m.w);
    }
  if (!gmax) throw std::runtime_error("pcs_open: no matrices");
  const unsigned log_gmax = log2_strict(gmax);
  PhaseTrace tr(ctx);

  // The usual pair of points of a matrix is (z, z * g), g the generator of its trace domain: the second point then needs no
  // inverse denominators of its own - 1 / (z g - x_j) = g^-1 / (z - x_sigma(j)) for a permutation sigma of the storage order
  // (open.hip::rev_dec) - which saves one full-domain pass per trace height (MSAMD_NO_NEXT_SHIFT=1: every point its own pass)
  const bool allow_next = !getenv("MSAMD_NO_NEXT_SHIFT");
  std::vector<std::vector<char>> is_next(rounds.size());
  for (size_t ri = 0; ri < rounds.size(); ri++) {
    auto& r = rounds[ri];
    is_next[ri].assign(r.data->ldes.size(), 0);
    for (size_t mi = 0; allow_next && mi < r.data->ldes.size(); mi++) {
      auto& pts = r.points[mi];
      const unsigned lh = log2_strict(r.data->ldes[mi].h);
      if (pts.size() != 2 || lh < lb || lh > 31) continue;
      const bool s0 = is_sym_point(pts[0]), s1 = is_sym_point(pts[1]);
      if (s0 && s1)
        is_next[ri][mi] = sym && pts[0].c0 == 0 && pts[1].c0 < sym->next_log.size() && sym->next_log[pts[1].c0] == (int)(lh - lb);
      else if (!s0 && !s1)
        is_next[ri][mi] = e2_same(pts[1], e2_mul_base(pts[0], gl_two_adic_generator(lh - lb))) && !e2_same(pts[0], pts[1]);
    }
  }
  // (a matrix of the same height that opens the second point in its own right would make it a third point of that height
  // for the reduced openings: keep the plain form there)
  for (size_t ri = 0; ri < rounds.size(); ri++)
    for (size_t mi = 0; mi < rounds[ri].data->ldes.size(); mi++) {
      if (!is_next[ri][mi]) continue;
      const size_t h = rounds[ri].data->ldes[mi].h;
      const E2 second = rounds[ri].points[mi][1];
      for (size_t rj = 0; rj < rounds.size() && is_next[ri][mi]; rj++)
        for (size_t mj = 0; mj < rounds[rj].data->ldes.size(); mj++) {
          if (rounds[rj].data->ldes[mj].h != h) continue;
          auto& q = rounds[rj].points[mj];
          for (size_t pj = 0; pj < q.size(); pj++)
            if (!(pj == 1 && is_next[rj][mj]) && e2_same(q[pj], second)) is_next[ri][mi] = 0;
        }
    }
  // unique opening points and the tallest matrix opened at each
  std::vector<E2> upts;
  std::vector<size_t> uh;
  auto point_index = [&](E2 z) -> size_t {
    for (size_t i = 0; i < upts.size(); i++)
      if (e2_same(upts[i], z)) return i;
    upts.push_back(z);
    uh.push_back(0);
    return upts.size() - 1;
  };
  for (size_t ri = 0; ri < rounds.size(); ri++) {
    auto& r = rounds[ri];
    for (size_t mi = 0; mi < r.data->ldes.size(); mi++)
      for (size_t pi = 0; pi < r.points[mi].size(); pi++) {
        if (pi == 1 && is_next[ri][mi]) continue;  // read through the first point's arrays
        size_t k = point_index(r.points[mi][pi]);
        uh[k] = std::max(uh[k], r.data->ldes[mi].h);
      }
  }
  // short matrices beside tall ones: their launches go to the side stream (see prove()), queued behind the tall ones'
  ctx.side_config();
  const size_t short_h = size_t(1) << (ctx.side_max_log + lb);
  const bool use_side = ctx.side_enabled && gmax > short_h && [&]() {
    for (auto& r : rounds)
      for (size_t mi = 0; mi < r.data->ldes.size(); mi++)
        if (r.data->ldes[mi].h <= short_h && !r.points[mi].empty()) return true;
    return false;
  }();
  struct SideSession {
    Ctx& c;
    ~SideSession() { c.side_join(); }
  } side_session{ctx};
  std::vector<DBuf<E2>> dens(upts.size()), xdens(upts.size());
  for (int pass = 0; pass < 2; pass++) {
    // the short matrices' sums also read the tall points' weights: fork once those are queued
    if (pass == 1 && use_side) ctx.side_fork();
    for (size_t k = 0; k < upts.size(); k++) {
      if ((int)(use_side && uh[k] <= short_h) != pass) continue;
      SideScope sc(ctx, pass == 1);
      dens[k] = DBuf<E2>(ctx, uh[k]);
      xdens[k] = DBuf<E2>(ctx, uh[k] >> lb);  // weights of the trace-domain coset: a prefix in bit-reversed storage
      if (is_sym_point(upts[k])) {
        if (!sym || upts[k].c0 >= sym->n) throw std::runtime_error("pcs_open: symbolic opening point without a device value");
        inv_denoms_dev(ctx, sym->d_points + upts[k].c0, log2_strict(uh[k]), dens[k].p, xdens[k].p, uh[k] >> lb);
      } else {
        inv_denoms(ctx, upts[k], log2_strict(uh[k]), dens[k].p, xdens[k].p, uh[k] >> lb);
      }
    }
  }

  tr.mark("inv_denoms");
  // opened values (barycentric over the first h = height / B storage rows): every matrix is launched first, one
  // read-back serves them all, then they are observed in round -> matrix -> point order
  opened.clear();
  size_t total_vals = 0;
  for (auto& r : rounds)
    for (size_t mi = 0; mi < r.data->ldes.size(); mi++) {
      if (r.points[mi].size() > 2) throw std::runtime_error("pcs_open: more than two points per matrix");
      total_vals += r.points[mi].size() * r.data->ldes[mi].w;
    }
  DBuf<E2> d_sums(ctx, std::max<size_t>(total_vals, 1));
  DBuf<E2> bary_partials[2];
  const bool batch_bary = !getenv("MSAMD_NO_BARY_BATCH");
  for (int pass = 0; pass < 2; pass++) {
    size_t off = 0;
    std::vector<BarySpec> specs;  // the matrices of one stream share a pair of launches
    for (size_t ri = 0; ri < rounds.size(); ri++) {
      auto& r = rounds[ri];
      for (size_t mi = 0; mi < r.data->ldes.size(); mi++) {
        const DMat& m = r.data->ldes[mi];
        auto& pts = r.points[mi];
        if (pts.empty()) continue;
        int np = (int)pts.size();
        if ((int)(use_side && m.h <= short_h) == pass) {
          const bool nx = is_next[ri][mi];
          const E2* d0 = xdens[point_index(pts[0])].p;
          const E2* d1 = np == 2 && !nx ? xdens[point_index(pts[1])].p : d0;
          if (batch_bary) {
            specs.push_back(BarySpec{m.d(), m.h, m.w, log2_strict(m.h) - lb, d0, d1, np, d_sums.p + off, nx});
          } else {
            SideScope sc(ctx, pass == 1);
            bary_sums_async(ctx, m.d(), m.h, m.w, log2_strict(m.h) - lb, d0, d1, np, d_sums.p + off, nx);
          }
        }
        off += np * m.w;
      }
    }
    if (!specs.empty()) {
      SideScope sc(ctx, pass == 1);
      bary_sums_batch(ctx, specs, bary_partials[pass]);
    }
  }
  std::vector<E2> h_sums(std::max<size_t>(total_vals, 1));
  g_probes.mark("opened values queued");
  // The opened values' transcript step on the DEVICE (open_alpha_k): with the outer transcript already there (`sym`) and FRI's
  // rounds device-driven, the finishing factors, the absorption of every opened value, the batching challenge alpha, its powers
  // and the reduced openings' coefficients and constants are one single-workgroup launch, and the proof's only wait is FRI's:
  // the raw sums arrive with it, and the host then replays the outer transcript, finishes the sums itself, absorbs the values
  // and compares its alpha with the device's. OPT-IN (MSAMD_DEV_OPENING=1): measured EQUAL to the host doing the step between
  // two waits (5.58-5.61 against 5.51-5.57 ms per HBM-resident proof, A / B on one box): the 64 + 13 us the GPU idles at the
  // opened-values wait are traded for a 25-30 us single-workgroup launch on the critical path and for the host's share of the
  // step (finishing the sums, absorbing the values: ~30 us) moving behind the proof's last kernel.
  const size_t fri_stop = (size_t(1) << lb) << prm.log_final_poly_len;
  size_t n_entry_pairs = 0;
  for (auto& r : rounds)
    for (auto& pts : r.points) n_entry_pairs += pts.size();
  const bool dev_open = sym && sym->d_state && gmax > fri_stop && prm.cap_height == 0 && prm.commit_pow_bits <= 16 && prm.max_log_arity == 1 &&
                        32 + 16 * total_vals <= (size_t(256) << 10) && gw < (size_t(1) << 20) && n_entry_pairs <= 512 && !getenv("MSAMD_HOST_FRI") && getenv("MSAMD_DEV_OPENING");
  auto finish_and_observe = [&]() {  // host: the opened values from the raw sums (points known), absorbed in round -> matrix -> point order
    if (sym) {  // the host learns the points now (and replays the transcript that produced them)
      std::vector<E2> values(sym->n);
      sym->resolve(values);
      auto swap_in = [&](E2& z) {
        if (is_sym_point(z)) z = values[z.c0];
      };
      for (auto& z : upts) swap_in(z);
      for (auto& r : rounds)
        for (auto& pts : r.points)
          for (auto& z : pts) swap_in(z);
    }
    size_t off = 0;
    for (auto& r : rounds) {
      OpenedRound orr;
      for (size_t mi = 0; mi < r.data->ldes.size(); mi++) {
        const DMat& m = r.data->ldes[mi];
        auto& pts = r.points[mi];
        std::vector<std::vector<E2>> per_point;
        if (!pts.empty()) {
          int np = (int)pts.size();
          std::vector<E2> ys(np * m.w);
          bary_finish(&h_sums[off], m.w, log2_strict(m.h) - lb, pts.data(), np, ys.data());
          off += np * m.w;
          for (int p = 0; p < np; p++) {
            per_point.emplace_back(ys.begin() + p * m.w, ys.begin() + (p + 1) * m.w);
            for (auto& y : per_point.back()) ch.observe_ext(y);
          }
        }
        orr.push_back(std::move(per_point));
      }
      opened.push_back(std::move(orr));
    }
  };
  E2 alpha = e2(0);
  std::vector<E2> apow(gw + 1);
  if (dev_open) {
    ctx.d2h_queue(h_sums.data(), d_sums.p, total_vals * sizeof(E2));
  } else {
    ctx.d2h(h_sums.data(), d_sums.p, total_vals * sizeof(E2));
    g_probes.mark("sync 4 (opened values)");
    finish_and_observe();
    tr.mark("bary_eval");
    g_probes.mark("opened values observed");
    alpha = ch.sample_ext();
    apow[0] = e2(1);
    for (size_t i = 1; i <= gw; i++) apow[i] = e2_mul(apow[i - 1], alpha);
  }
  // reduced openings per LDE height; the opening points of one height are numbered locally (at most two). Everything but the
  // coefficients and the constants K follows from the opening's SHAPE; those two come from alpha and the opened values - on the
  // host here, or from open_alpha_k, which gets one OpenEntry per (matrix, point) saying where its inputs and outputs lie
  std::vector<size_t> num_reduced(33, 0);
  std::vector<std::vector<DeepMat>> lists(33);
  std::vector<DeepPoints> hpts(33);
  std::vector<std::vector<size_t>> hpt_global(33);
  std::vector<char> present(33, 0);
  std::vector<OpenEntry> entries;                        // dev_open: observe order
  std::vector<std::pair<unsigned, size_t>> entry_mat;    // (height, index in lists[height]) of each entry's matrix
  constexpr size_t NEXT_MARK = size_t(1) << 62;
  for (auto& hp : hpts) memset(&hp, 0, sizeof(hp));
  {
    size_t sum_off = 0, out_off = 0;
    for (size_t ri = 0; ri < rounds.size(); ri++) {
      auto& r = rounds[ri];
      for (size_t mi = 0; mi < r.data->ldes.size(); mi++) {
        const DMat& m = r.data->ldes[mi];
        unsigned lh = log2_strict(m.h);
        present[lh] = 1;
        auto& pts = r.points[mi];
        if (pts.empty()) continue;
        DeepMat dm;
        memset(&dm, 0, sizeof(dm));
        dm.d = m.d();
        dm.w = (uint32_t)m.w;
        dm.npoints = (uint32_t)pts.size();
        for (size_t pi = 0; pi < pts.size(); pi++) {
          // a "next" point is named by the first point's arrays plus a mark (all matrices of one height share g, so the pair
          // (arrays, mark) identifies the point at this height)
          const bool nx = pi == 1 && is_next[ri][mi];
          const size_t gk = point_index(pts[nx ? 0 : pi]) | (nx ? NEXT_MARK : size_t(0));
          size_t local = 0;
          while (local < hpt_global[lh].size() && hpt_global[lh][local] != gk) local++;
          if (local == hpt_global[lh].size()) {
            if (local == 2) throw std::runtime_error("pcs_open: more than two opening points at one LDE height");
            hpt_global[lh].push_back(gk);
            hpts[lh].den[local] = dens[gk & ~NEXT_MARK].p;
            hpts[lh].shift[local] = nx ? (uint32_t(1) << lb) : 0u;  // g = w_H^blowup
            hpts[lh].K[local] = e2(0);
            hpts[lh].n = (uint32_t)(local + 1);
          }
          const u64 cmul = nx ? gl_inv(gl_two_adic_generator(lh - lb)) : 1;  // 1 / (z g - x_j) = g^-1 / (z - x_sigma(j))
          dm.pt[pi] = (uint32_t)local;
          if (dev_open) {
            const unsigned log_h = lh - lb;
            const u64 s_pow = gl_exp_pow2(GL_GEN, log_h);
            OpenEntry en;
            memset(&en, 0, sizeof(en));
            en.sum_off = (uint32_t)sum_off;
            en.out_off = (uint32_t)out_off;
            en.w = (uint32_t)m.w;
            en.np = (uint32_t)pts.size();
            en.p = (uint32_t)pi;
            en.log_h = log_h;
            if (!is_sym_point(pts[pi])) throw std::runtime_error("pcs_open: device opening needs device points");
            en.point_id = (uint32_t)pts[pi].c0;
            en.exp = (uint32_t)num_reduced[lh];
            en.slot = (uint32_t)(2 * lh + local);
            en.s_pow = s_pow;
            en.dinv = gl_inv(gl_mul(s_pow, (u64(1) << log_h) % GL_P));
            en.cmul = cmul;
            entries.push_back(en);
            entry_mat.emplace_back(lh, lists[lh].size());
          } else {
            E2 coeff = e2_pow(alpha, num_reduced[lh]);
            E2 rz = e2(0);
            const std::vector<E2>& ys = opened[ri][mi][pi];
            for (size_t c = 0; c < m.w; c++) rz = e2_add(rz, e2_mul(apow[c], ys[c]));
            coeff = e2_mul_base(coeff, cmul);
            dm.coeff[pi] = coeff;
            dm.coeff7[pi] = gl_mul(coeff.c1, GL_EXT_W);
            hpts[lh].K[local] = e2_add(hpts[lh].K[local], e2_mul(coeff, rz));
          }
          num_reduced[lh] += m.w;
          out_off += m.w;
        }
        sum_off += pts.size() * m.w;
        lists[lh].push_back(dm);
      }
    }
  }
  g_probes.mark("reduced-opening coefficients");
  // the alpha powers and every height's matrix list cross in ONE copy (each copy is a launch of its own in the stream)
  static_assert(sizeof(DeepMat) % 8 == 0 && sizeof(E2) == 16, "the blob keeps both aligned");
  size_t n_mats = 0;
  for (auto& l : lists) n_mats += l.size();
  std::vector<uint8_t> deep_blob((gw + 1) * sizeof(E2) + n_mats * sizeof(DeepMat));
  if (!dev_open) memcpy(deep_blob.data(), apow.data(), (gw + 1) * sizeof(E2));
  std::vector<size_t> list_off(33, 0);
  {
    size_t off = (gw + 1) * sizeof(E2);
    for (int lh = 32; lh >= 0; lh--) {
      list_off[lh] = off;
      if (lists[lh].empty()) continue;
      memcpy(deep_blob.data() + off, lists[lh].data(), lists[lh].size() * sizeof(DeepMat));
      off += lists[lh].size() * sizeof(DeepMat);
    }
  }
  DBuf<uint8_t> d_deep(ctx, deep_blob.size());
  ctx.h2d(d_deep.p, deep_blob.data(), deep_blob.size());
  const E2* d_apow = reinterpret_cast<const E2*>(d_deep.p);
  DBuf<E2> d_K, d_opened, d_alpha;
  DBuf<uint32_t> d_fri_state;
  DBuf<OpenEntry> d_entries;
  DBuf<Digest> d_cvs;
  E2 h_alpha_dev = e2(0);
  if (dev_open) {
    for (size_t e = 0; e < entries.size(); e++)
      entries[e].mat = (uint32_t)((list_off[entry_mat[e].first] - (gw + 1) * sizeof(E2)) / sizeof(DeepMat) + entry_mat[e].second);
    d_entries = DBuf<OpenEntry>(ctx, std::max<size_t>(entries.size(), 1));
    ctx.h2d(d_entries.p, entries.data(), entries.size() * sizeof(OpenEntry));
    d_K = DBuf<E2>(ctx, 66);
    d_opened = DBuf<E2>(ctx, std::max<size_t>(total_vals, 1));
    d_alpha = DBuf<E2>(ctx, 1);
    d_fri_state = DBuf<uint32_t>(ctx, 8);
    d_cvs = DBuf<Digest>(ctx, (32 + 16 * total_vals + 1023) / 1024);
    OpenAlphaArgs oa;
    memset(&oa, 0, sizeof(oa));
    oa.entries = d_entries.p;
    oa.n_entries = (uint32_t)entries.size();
    oa.n_vals = (uint32_t)total_vals;
    oa.gw = (uint32_t)gw;
    oa.n_slots = 66;
    oa.sums = d_sums.p;
    oa.points = sym->d_points;
    oa.state_in = sym->d_state;
    oa.opened = d_opened.p;
    oa.apow = reinterpret_cast<E2*>(d_deep.p);
    oa.mats = reinterpret_cast<DeepMat*>(d_deep.p + (gw + 1) * sizeof(E2));
    oa.K = d_K.p;
    oa.state_out = d_fri_state.p;
    oa.alpha_out = d_alpha.p;
    oa.cv_scratch = d_cvs.p;
    open_alpha(ctx, oa);
    ctx.d2h_queue(&h_alpha_dev, d_alpha.p, sizeof(E2));
  }
  std::vector<DBuf<E2>> inputs;  // descending height
  DTree fri_round0;              // the tallest vector is FRI's first committed matrix: its leaf layer is hashed where it is produced
  if (use_side) ctx.side_fork();  // behind the upload of the alpha powers (and the launch that fills them)
  for (int lh = 32; lh >= 0; lh--) {
    if (!present[lh]) continue;
    size_t h = size_t(1) << lh;
    SideScope sc(ctx, use_side && h <= short_h);
    DBuf<E2> ro(ctx, h);
    if (lists[lh].empty()) {
      HIP_CHECK(hipMemsetAsync(ro.p, 0, h * sizeof(E2), ctx.stream));
    } else {
      Digest* leaves = nullptr;
      if (inputs.empty() && h >= 8192 && prm.max_log_arity == 1 && !getenv("MSAMD_NO_DEEP_LEAVES")) {
        merkle_alloc(ctx, fri_round0, h / 2);
        leaves = fri_round0.base();
      }
      deep_reduce(ctx, lists[lh], hpts[lh], h, d_apow, ro.p, apow.data(), leaves, reinterpret_cast<const DeepMat*>(d_deep.p + list_off[lh]), 0, 0,
                  dev_open ? d_K.p + 2 * lh : nullptr);
    }
    inputs.push_back(std::move(ro));
  }
  g_probes.mark("reduced openings queued");
  ctx.side_join();  // the short reduced openings read the tall points' denominators, released next
  for (auto& d : dens) d.reset();
  for (auto& d : xdens) d.reset();
  tr.mark("deep_reduce");

  // ---- query-phase segments of the input rounds (what to read for ONE query), then FRI itself
  InputShape shape;
  std::vector<GatherSeg> segs;
  size_t out_off = 0;
  for (auto& r : rounds) {
    const DTree& t = r.data->tree;
    const unsigned lmh = log2_strict(t.max_height());
    const unsigned sh0 = log_gmax - lmh;
    std::vector<size_t> widths;
    for (auto& m : r.data->ldes) {
      add_gather_seg(segs, out_off, m.d(), m.h, (uint32_t)m.w, 0, sh0 + (lmh - log2_strict(m.h)), 0);
      widths.push_back(m.w);
    }
    for (size_t i = 0; i < t.cap_layer(); i++) add_gather_seg(segs, out_off, t.base() + t.layer_off[i], 0, 1, 1, sh0 + (unsigned)i, 1);
    shape.widths.push_back(std::move(widths));
    shape.nsib.push_back(t.cap_layer());
  }
  FriDeviceStart dstart;
  if (dev_open) {
    dstart.d_state = std::move(d_fri_state);
    dstart.after_wait = [&]() {  // FRI's wait has delivered the raw sums, the commitments and the device's challenges
      finish_and_observe();
      alpha = ch.sample_ext();
      if (!e2_same(alpha, h_alpha_dev)) throw std::runtime_error("the device transcript's batching challenge differs from the host challenger's");
      g_probes.mark("opened values observed");
    };
  }
  fri_prove(sys, ch, inputs, log_gmax, segs, out_off, shape, nullptr, fri_bytes, tr, nullptr, &fri_round0, dev_open ? &dstart : nullptr);
}

// prove_fri (commit phase, final polynomial, query proof of work, query openings) over the reduced openings
// `inputs` (descending height); serialises the FriProof into `fri_bytes`. The openings of the INPUT rounds are read
// either by `input_segs` on this device (their per-query block of `input_qbytes` bytes laid out round by round:
// every matrix row, then the sibling digests bottom-up) or, when the committed data is spread over ranks, by
// `remote`, called with the sampled indices and filling the same layout.
void fri_prove(HSystem& sys, Challenger& ch, std::vector<DBuf<E2>>& inputs, unsigned log_gmax, const std::vector<GatherSeg>& input_segs,
               size_t input_qbytes, const InputShape& shape, const InputGather* remote, PW& fri_bytes, PhaseTrace& tr, FriHead* head,
               DTree* round0, FriDeviceStart* dstart) {
  Ctx& ctx = *sys.ctx;
  const unsigned head_rounds = head ? head->n_rounds : 0;
  if (head && !remote) throw std::runtime_error("FRI: head rounds need the remote gather");
  const Params& prm = sys.params;
  const unsigned lb = (unsigned)prm.log_blowup;
  (void)log_gmax;
  // ---- FRI commit phase (prove_fri / commit_phase)
  const size_t final_len = size_t(1) << prm.log_final_poly_len;
  const size_t stop = (size_t(1) << lb) * final_len;
  // p3-fri's prove_fri asserts, when the final polynomial has more than one coefficient, that even the shortest input
  // is taller than blowup * final length: the reference panics there, this returns an error
  if (prm.log_final_poly_len > 0)
    for (auto& in : inputs)
      if (in.n <= stop) throw std::runtime_error("FRI: a committed matrix is not taller than blowup * final polynomial length");
  std::vector<DBuf<E2>> layer_bufs;   // owners of the folded vectors kept for the query phase
  std::vector<const E2*> layers;      // input vector of every commit-phase round
  std::vector<DTree> trees;
  std::vector<unsigned> arities;      // log2 of every round's arity (1 on every path but the host-driven one)
  std::vector<std::vector<Digest>> commits;
  std::vector<u64> pow_w;
  DBuf<E2> folded = std::move(inputs[0]);
  size_t next_in = 1;
  const unsigned log_max_height = head ? head->log_max_height : log2_strict(folded.n);
  std::vector<E2> fin;                // final folded vector (host)
  DBuf<Digest> tail_tree;
  DBuf<E2> tail_layers;
  // ---- device transcript: with a root-only commitment and the challenger's pending input being exactly one digest
  // (true right after the alpha sample) every round's challenger step runs on the device (challenge_dev.h), so the
  // whole commit phase is submitted without a host synchronisation. The host then replays the transcript from
  // the returned roots / witnesses on its own challenger; the device values are checked, not trusted.
  // Rounds of arity above 2 (max_log_arity > 1; no call site of the reference folds wider, src/types.rs:189-190) take the same
  // device transcript since the end of round 4, one round at a time: wide leaves + tree + challenger step in the launches of
  // fri_tree_build, then the round's binary folds with beta, beta^2, beta^4 .. read from the round's record; no fused rounds,
  // no single-workgroup tail (MSAMD_HOST_WIDE_FRI=1: host-driven, one synchronisation per round, as before).
  const bool wide = prm.max_log_arity > 1;
  const bool dev_rounds = folded.n > stop && prm.cap_height == 0 && (dstart || ch.input.size() == 32) && prm.commit_pow_bits <= 16 &&
                          (!wide || (!head && !getenv("MSAMD_HOST_WIDE_FRI"))) && !getenv("MSAMD_HOST_FRI");
  const unsigned log_final_height = lb + (unsigned)prm.log_final_poly_len;
  // p3-fri compute_log_arity_for_round: as far as max_log_arity allows without stepping over the next input or the final height
  auto round_arity = [&](size_t n, size_t ni) {
    const unsigned lh = log2_strict(n);
    unsigned la = std::min<unsigned>((unsigned)prm.max_log_arity, lh - log_final_height);
    if (ni < inputs.size()) la = std::min(la, lh - log2_strict(inputs[ni].n));
    if (la < 1) throw std::runtime_error("FRI: two inputs of one height");
    return la;
  };
  if (dstart && !dev_rounds) throw std::runtime_error("FRI: a transcript that starts on the device needs device-driven rounds");
  // With a one-coefficient final polynomial the query phase's challenger work (observe the final polynomial, grind,
  // sample every index) also runs on the device and the openings are gathered from the device-side indices, so
  // the whole of FRI costs one host synchronisation; the host replay below checks witness and indices.
  if (head && head->on_device && !dev_rounds) throw std::runtime_error("FRI: head rounds on the device need the device transcript for the rest");
  const bool dev_query = dev_rounds && final_len == 1 && prm.query_pow_bits <= 16 && prm.num_queries <= 4096 && !getenv("MSAMD_HOST_QUERY");
  size_t n_total = 0;
  DBuf<uint32_t> d_state;
  DBuf<FriTailRound> d_recs;
  DBuf<E2> d_final, fin_hold;
  const E2* fin_src = nullptr;
  if (dev_rounds) {
    const bool use_tail = !getenv("MSAMD_NO_FRI_TAIL");
    if (wide) {
      size_t ni = next_in;
      for (size_t l = folded.n; l > stop;) {
        l >>= round_arity(l, ni);
        if (ni < inputs.size() && inputs[ni].n == l) ni++;
        n_total++;
      }
    } else {
      for (size_t l = folded.n; l > stop; l >>= 1) n_total++;
    }
    if (dstart) {
      d_state = std::move(dstart->d_state);  // the opened values were absorbed on the device: FRI goes on from that state
    } else if (head && head->on_device) {
      d_state = std::move(head->d_state);  // the head rounds' challenger steps have run on the device: go on from their state
    } else {
      d_state = DBuf<uint32_t>(ctx, 8);
      ctx.h2d(d_state.p, ch.input.data(), 32);
    }
    d_recs = DBuf<FriTailRound>(ctx, n_total);
    d_final = DBuf<E2>(ctx, stop);
    size_t r = 0;
    bool leaves_done = false;  // the fold of the previous round already hashed this round's leaves
    bool tree_done = false;    // the previous round's fused launch already built this round's tree and ran its challenger step
    if (round0 && round0->digests.p && folded.n > stop && !(use_tail && folded.n <= 2048)) {
      // the first round's leaf layer was hashed by the pass that produced the vector (deep_reduce_k)
      round0->cap_height = 0;
      trees.push_back(std::move(*round0));
      leaves_done = true;
    }
    while (folded.n > stop) {
      if (wide) {
        const unsigned la = round_arity(folded.n, next_in);
        const size_t rows = folded.n >> la;
        trees.emplace_back();
        trees.back().cap_height = 0;
        FriChallenge fc{d_state.p, d_recs.p + r, (uint32_t)prm.commit_pow_bits};
        fri_tree_build(ctx, trees.back(), folded.p, rows, &fc, la);
        const E2* roll = nullptr;
        if (next_in < inputs.size() && inputs[next_in].n == rows) roll = inputs[next_in++].p;
        const E2* src = folded.p;
        DBuf<E2> step;
        for (unsigned j = 0; j < la; j++) {
          const size_t out_rows = folded.n >> (j + 1);
          DBuf<E2> nxt(ctx, out_rows);
          fri_fold_dev(ctx, src, out_rows, d_recs.p + r, j + 1 == la ? roll : nullptr, nxt.p, nullptr, 0, 0, j);
          step = std::move(nxt);  // (the previous intermediate vector is released behind the launch that read it: stream-ordered pool)
          src = step.p;
        }
        layers.push_back(folded.p);
        arities.push_back(la);
        layer_bufs.push_back(std::move(folded));
        folded = std::move(step);
        r++;
        continue;
      }
      if (use_tail && folded.n <= 2048) {
        const uint32_t len0 = (uint32_t)folded.n;
        const uint32_t n_rounds = (uint32_t)(n_total - r);
        size_t tree_digests = 0, layer_elems = 0;
        for (size_t l = len0; l > stop; l >>= 1) {
          tree_digests += l - 1;            // rows + rows/2 + ... + 1 with rows = l/2
          if (l != len0) layer_elems += l;  // inputs of rounds 1..
        }
        std::vector<FriTailRoll> rolls;
        for (size_t k = next_in; k < inputs.size(); k++) rolls.push_back(FriTailRoll{inputs[k].p, (uint32_t)inputs[k].n, 0});
        tail_tree = DBuf<Digest>(ctx, tree_digests);
        tail_layers = DBuf<E2>(ctx, std::max<size_t>(layer_elems, 1));
        fri_tail(ctx, folded.p, len0, n_rounds, (unsigned)prm.commit_pow_bits, d_state.p, rolls, tail_tree.p, tail_layers.p, d_recs.p + r,
                 d_final.p);
        size_t toff = 0, loff = 0, l = len0;
        for (uint32_t k = 0; k < n_rounds; k++, l >>= 1) {
          const size_t rows = l / 2;
          trees.emplace_back();
          DTree& t = trees.back();
          t.cap_height = 0;
          t.ext = tail_tree.p + toff;
          size_t o = 0;
          for (size_t n = rows;; n >>= 1) {
            t.layer_off.push_back(o);
            t.layer_len.push_back(n);
            o += n;
            if (n == 1) break;
          }
          toff += o;
          layers.push_back(k == 0 ? folded.p : tail_layers.p + loff);
          if (k > 0) loff += l;
          if (next_in < inputs.size() && inputs[next_in].n == rows) next_in++;
        }
        r += n_rounds;
        layer_bufs.push_back(std::move(folded));
        folded = DBuf<E2>();
        fin_src = d_final.p;
        break;
      }
      const size_t rows = folded.n / 2;
      if (!tree_done) {
        if (!leaves_done) {
          trees.emplace_back();
          trees.back().cap_height = 0;
        }
        FriChallenge fc{d_state.p, d_recs.p + r, (uint32_t)prm.commit_pow_bits};
        fri_tree_build(ctx, trees.back(), leaves_done ? nullptr : folded.p, rows, &fc);
      }
      DBuf<E2> nxt(ctx, rows);
      const E2* roll = nullptr;
      if (next_in < inputs.size() && inputs[next_in].n == rows) roll = inputs[next_in++].p;
      // does the folded vector get a commit round of its own outside the tail kernel?
      const bool next_is_round = rows > stop && rows >= 2 && !(use_tail && rows <= 2048);
      if (next_is_round && fri_round_fusable(rows)) {
        // that whole round - this fold, its leaf digests, its tree, its challenger step - is ONE launch
        trees.emplace_back();
        trees.back().cap_height = 0;
        FriChallenge fc_next{d_state.p, d_recs.p + r + 1, (uint32_t)prm.commit_pow_bits};
        fri_round_fused(ctx, trees.back(), folded.p, rows, d_recs.p + r, roll, nxt.p, fc_next);
        tree_done = true;
        leaves_done = false;
      } else {
        tree_done = false;
        // the next round's leaf layer is hashed by the fold itself unless that round belongs to the tail kernel
        leaves_done = next_is_round;
        Digest* next_leaves = nullptr;
        if (leaves_done) {
          trees.emplace_back();
          trees.back().cap_height = 0;
          merkle_alloc(ctx, trees.back(), rows / 2);
          next_leaves = trees.back().base();
        }
        fri_fold_dev(ctx, folded.p, rows, d_recs.p + r, roll, nxt.p, next_leaves);
      }
      layers.push_back(folded.p);
      layer_bufs.push_back(std::move(folded));
      folded = std::move(nxt);
      r++;
    }
    if (r != n_total) throw std::runtime_error("FRI: round count mismatch");
    if (!fin_src) {
      fin_src = folded.p;
      fin_hold = std::move(folded);
    }
    folded = DBuf<E2>();
  }
  while (folded.n > stop) {  // host-driven rounds (cap_height > 0, wide grinding, MSAMD_HOST_FRI / MSAMD_HOST_WIDE_FRI, a wide joint proof)
    const unsigned la = round_arity(folded.n, next_in);
    size_t rows = folded.n >> la;
    const bool prehashed = la == 1 && round0 && round0->digests.p && trees.empty() && layers.empty() && round0->layer_len[0] == rows;
    if (prehashed)
      trees.push_back(std::move(*round0));
    else
      trees.emplace_back();
    DTree& t = trees.back();
    t.cap_height = (unsigned)prm.cap_height;
    fri_tree_build(ctx, t, prehashed ? nullptr : folded.p, rows, nullptr, la);
    bool found = false;
    u64 wit = 0;
    std::vector<Digest> cap = cap_and_grind(ctx, t, ch.input, (unsigned)prm.commit_pow_bits, &found, &wit);
    ch.observe_cap(cap);
    commits.push_back(cap);
    if (found) {
      ch.observe(wit);
      if (ch.sample_bits((unsigned)prm.commit_pow_bits) != 0) throw std::runtime_error("grind: device witness rejected by the host challenger");
      pow_w.push_back(wit);
    } else {
      pow_w.push_back(grind(ctx, ch, (unsigned)prm.commit_pow_bits));
    }
    E2 beta = ch.sample_ext();
    // a round of arity 2^la is la binary folds with beta, beta^2, beta^4, ... (the value at beta of the polynomial of degree
    // < 2^la through a row); the vector rolled in behind it takes beta^(2^la) - the square of the last step's challenge
    const E2* roll = nullptr;
    if (next_in < inputs.size() && inputs[next_in].n == rows) roll = inputs[next_in].p;
    const E2* src = folded.p;
    DBuf<E2> step;
    for (unsigned j = 0; j < la; j++) {
      const size_t out_rows = folded.n >> (j + 1);
      DBuf<E2> nxt(ctx, out_rows);
      fri_fold(ctx, src, out_rows, beta, j + 1 == la ? roll : nullptr, nxt.p);
      beta = e2_sqr(beta);
      step = std::move(nxt);  // (the previous intermediate vector is released behind the launch that read it: stream-ordered pool)
      src = step.p;
    }
    if (roll) next_in++;
    layers.push_back(folded.p);
    arities.push_back(la);
    layer_bufs.push_back(std::move(folded));
    folded = std::move(step);
  }
  if (next_in != inputs.size()) throw std::runtime_error("FRI: an input was never rolled in");

  // ---- query phase, part 1: the segment list (what to read for ONE query) needs no challenge
  std::vector<GatherSeg> segs;
  size_t out_off = 0;
  if (!remote) {
    segs = input_segs;
    out_off = input_qbytes;
  }
  auto n_siblings = [](const DTree& t) { return t.cap_layer(); };
  arities.resize(trees.size(), 1);
  {
    uint32_t gr = head_rounds;  // index bits consumed by the rounds before this one (rounds count from the head's)
    for (size_t i = 0; i < trees.size(); i++) {
      const unsigned la = arities[i];
      // binary round: the sibling value sits at element (index >> gr) ^ 1 of the round's vector; wider: the whole row
      // (index >> gr) >> la is fetched and the queried position's own value dropped when the bytes are written
      if (la == 1)
        add_gather_seg(segs, out_off, layers[i], 0, 1, 2, gr, 1);
      else
        add_gather_seg(segs, out_off, layers[i], 0, 2u << la, 3, gr + la, 0);
      const DTree& t = trees[i];
      for (size_t l = 0; l < n_siblings(t); l++) add_gather_seg(segs, out_off, t.base() + t.layer_off[l], 0, 1, 1, (uint32_t)(gr + la + l), 1);
      gr += la;
    }
  }
  const size_t qbytes = out_off;
  const size_t nq = (size_t)prm.num_queries;
  // the gathered openings (about 1 MB at the bench parameters) are read where the read-back lands - the pinned staging
  // buffer - when they fit there; otherwise, and for the host-driven gathers, in a scratch vector that is allocated once
  std::vector<uint8_t>& g_vec = ctx.host_scratch;
  const uint8_t* g = nullptr;
  std::vector<u64> dq(1 + nq);  // device query step: [0] = PoW witness, [1..] = indices
  if (dev_rounds) {
    DBuf<u64> d_q;
    DBuf<uint8_t> d_g;
    DBuf<GatherSeg> d_segs;
    if (dev_query) {
      d_q = DBuf<u64>(ctx, 1 + nq);
      fri_query_challenge(ctx, d_state.p, fin_src, (unsigned)prm.query_pow_bits, (uint32_t)nq, log_max_height, d_q.p);
      d_g = DBuf<uint8_t>(ctx, std::max<size_t>(qbytes * nq, 1));
      d_segs = DBuf<GatherSeg>(ctx, std::max<size_t>(segs.size(), 1));
      gather_queries_launch(ctx, segs, d_segs.p, d_q.p + 1, nq, qbytes, d_g.p);
      if (remote && remote->queue) remote->queue(d_q.p + 1, nq);  // the input rounds' openings travel with this synchronisation too
      ctx.d2h_queue(dq.data(), d_q.p, dq.size() * 8);
      // (not when the input openings are fetched from other ranks afterwards: those read-backs reuse the staging buffer)
      g = remote ? nullptr : ctx.d2h_queue_staged(d_g.p, qbytes * nq);
      if (!g) {
        if (g_vec.size() < qbytes * nq) g_vec.resize(qbytes * nq);
        ctx.d2h_queue(g_vec.data(), d_g.p, qbytes * nq);
        g = g_vec.data();
      }
    }
    std::vector<FriTailRound> recs(n_total), hrecs(head && head->on_device ? head_rounds : 0);
    fin.resize(stop);
    ctx.d2h_queue(recs.data(), d_recs.p, n_total * sizeof(FriTailRound));
    if (!hrecs.empty()) ctx.d2h_queue(hrecs.data(), head->d_recs.p, hrecs.size() * sizeof(FriTailRound));
    g_probes.mark("FRI queued");
    ctx.d2h(fin.data(), fin_src, stop * sizeof(E2));  // the one synchronisation of the FRI phase
    g_probes.mark("sync 5 (FRI)");
    fin_hold.reset();
    if (dstart && dstart->after_wait) dstart->after_wait();  // the host catches up with what the device did in front of FRI
    for (size_t k = 0; k < hrecs.size(); k++) {  // the rounds that ran on row shards, replayed first: they come first in the transcript
      Digest root;
      memcpy(root.b, hrecs[k].root, 32);
      std::vector<Digest> cap(1, root);
      ch.observe_cap(cap);
      head->commits.push_back(cap);
      if (prm.commit_pow_bits) {
        ch.observe(hrecs[k].witness);
        if (ch.sample_bits((unsigned)prm.commit_pow_bits) != 0) throw std::runtime_error("FRI: device witness rejected by the host challenger");
      }
      head->pow.push_back(prm.commit_pow_bits ? hrecs[k].witness : 0);
      if (!e2_same(ch.sample_ext(), hrecs[k].beta)) throw std::runtime_error("FRI: device challenger diverged from the host transcript");
    }
    for (size_t k = 0; k < n_total; k++) {
      Digest root;
      memcpy(root.b, recs[k].root, 32);
      std::vector<Digest> cap(1, root);
      ch.observe_cap(cap);
      commits.push_back(cap);
      if (prm.commit_pow_bits) {
        ch.observe(recs[k].witness);
        if (ch.sample_bits((unsigned)prm.commit_pow_bits) != 0) throw std::runtime_error("FRI: device witness rejected by the host challenger");
        // minimality (the reference accepts any witness; ours is pinned to the smallest) is the kernel's atomicMin
      }
      pow_w.push_back(prm.commit_pow_bits ? recs[k].witness : 0);
      E2 beta = ch.sample_ext();
      if (!e2_same(beta, recs[k].beta)) throw std::runtime_error("FRI: device challenger diverged from the host transcript");
    }
  } else if (folded.p) {
    fin.resize(folded.n);
    ctx.d2h(fin.data(), folded.p, folded.n * sizeof(E2));
  }
  tr.mark("fri_commit_phase");
  g_probes.mark("FRI rounds replayed");
  const bool host_trace = getenv("MSAMD_TRACE_HOST") != nullptr;
  const double t_sync = host_trace ? now_ms() : 0;
  // final polynomial: truncate, undo the bit reversal, inverse DFT (tiny: on the host)
  if (fin.size() != stop) throw std::runtime_error("FRI: unexpected final length");
  std::vector<E2> final_poly(final_len);
  {
    unsigned lf = (unsigned)prm.log_final_poly_len;
    std::vector<E2> ev(final_len);
    for (size_t i = 0; i < final_len; i++) ev[bitrev64(i, lf)] = fin[i];
    u64 winv = gl_inv(gl_two_adic_generator(lf)), ninv = gl_inv((u64)final_len);
    for (size_t k = 0; k < final_len; k++) {
      E2 s = e2(0);
      u64 wk = gl_pow(winv, k), cur = 1;
      for (size_t j = 0; j < final_len; j++) {
        s = e2_add(s, e2_mul_base(ev[j], cur));
        cur = gl_mul(cur, wk);
      }
      final_poly[k] = e2_mul_base(s, ninv);
      ch.observe_ext(final_poly[k]);
    }
  }
  // ---- query phase, part 2: proof of work, indices, openings
  u64 query_pow = 0;
  std::vector<uint64_t> indices(nq);
  if (dev_query) {
    query_pow = dq[0];
    if (prm.query_pow_bits) {
      ch.observe(query_pow);
      if (ch.sample_bits((unsigned)prm.query_pow_bits) != 0) throw std::runtime_error("FRI: device query witness rejected by the host challenger");
    }
    for (size_t i = 0; i < nq; i++) {
      indices[i] = ch.sample_bits(log_max_height);
      if (indices[i] != dq[1 + i]) throw std::runtime_error("FRI: device query indices diverged from the host transcript");
    }
    tr.mark("final_poly+grind+gather");
    g_probes.mark("query indices replayed");
  } else {
    query_pow = grind(ctx, ch, (unsigned)prm.query_pow_bits);
    for (auto& ix : indices) ix = ch.sample_bits(log_max_height);
    tr.mark("final_poly+grind");
    if (g_vec.size() < qbytes * nq) g_vec.resize(qbytes * nq);
    gather_queries(ctx, segs, indices, qbytes, g_vec.data());
    g = g_vec.data();
    tr.mark("query_gather");
  }

  // ---- openings of the input rounds held elsewhere
  std::vector<uint8_t> input_g;
  const size_t remote_qbytes = input_qbytes + (head ? head->qbytes() : 0);
  if (remote) {
    input_g.resize(remote_qbytes * nq);
    remote->fetch(indices, remote_qbytes, input_g.data());
    tr.mark("remote_input_openings");
  }

  // ---- FriProof bytes
  PW& w = fri_bytes;
  w.b.reserve(w.b.size() + nq * (qbytes + remote_qbytes + 64 * (trees.size() + head_rounds + shape.widths.size() * 4 + 8)) + 4096);
  w.u64_(commits.size() + head_rounds);
  if (head)
    for (auto& c : head->commits) w.cap(c);
  for (auto& c : commits) w.cap(c);
  w.u64_(pow_w.size() + head_rounds);
  if (head)
    for (u64 x : head->pow) w.u64_(x);
  for (u64 x : pow_w) w.u64_(x);
  w.u64_(indices.size());
  size_t pos = 0;
  for (size_t qi = 0; qi < indices.size(); qi++) {
    const uint8_t* ip = remote ? &input_g[qi * remote_qbytes] : &g[pos];
    w.u64_(shape.widths.size());
    for (size_t ri = 0; ri < shape.widths.size(); ri++) {
      w.u64_(shape.widths[ri].size());
      for (size_t mw : shape.widths[ri]) {
        w.u64_(mw);
        w.raw(ip, mw * 8);
        ip += mw * 8;
      }
      const size_t ns = shape.nsib[ri];
      w.u64_(ns);
      w.raw(ip, ns * 32);
      ip += ns * 32;
    }
    if (!remote) pos += input_qbytes;
    w.u64_(trees.size() + head_rounds);
    for (unsigned hr = 0; hr < head_rounds; hr++) {  // rounds that ran on row shards: bytes fetched from the owning rank
      w.u8(1);  // log_arity
      w.u64_(1);
      w.raw(ip, 16);
      ip += 16;
      w.u64_(head->nsib[hr]);
      w.raw(ip, head->nsib[hr] * 32);
      ip += head->nsib[hr] * 32;
    }
    uint64_t index_i = indices[qi] >> head_rounds;
    for (size_t i = 0; i < trees.size(); i++) {
      const unsigned la = arities[i];
      w.u8((uint8_t)la);  // log_arity
      w.u64_((size_t(1) << la) - 1);
      if (la == 1) {
        w.raw(&g[pos], 16);
        pos += 16;
      } else {  // the row without the queried position's own value
        const size_t own = index_i & ((size_t(1) << la) - 1);
        w.raw(&g[pos], own * 16);
        w.raw(&g[pos + (own + 1) * 16], ((size_t(1) << la) - 1 - own) * 16);
        pos += size_t(16) << la;
      }
      index_i >>= la;
      size_t ns = n_siblings(trees[i]);
      w.u64_(ns);
      w.raw(&g[pos], ns * 32);
      pos += ns * 32;
    }
  }
  w.u64_(final_poly.size());
  for (auto& e : final_poly) w.ext(e);
  w.u64_(query_pow);
  g_probes.mark("FriProof bytes written");
  if (host_trace) fprintf(stderr, "[msamd] host work after the FRI read-back: %.1f us (transcript replay + FriProof bytes)\n", 1e3 * (now_ms() - t_sync));
}
}  // namespace

// beta / gamma in device memory (uploaded by the host or sampled there by outer.hip)
void stage2_circuit_dyn(Ctx& ctx, const HSystem& sys, const HWitness& wit, size_t ci, const ChallengeBG* bg, u64* out, E2* total_dev) {
  const HCircuit& c = sys.circuits[ci];
  const DLookups& lk = wit.lookups[ci];
  if (c.num_lookups && !lk.mult.p) {  // no LookupValues were materialised for this circuit: evaluate them in the kernel
    if (!c.stage2_trace_jit.function || !wit.traces[ci].p) throw std::runtime_error("stage 2: this circuit has neither lookup values nor a fused kernel");
    stage2_from_trace_dyn(ctx, c.stage2_trace_jit, wit.traces[ci].p, c.pre_width ? c.d_preprocessed.p : nullptr, wit.heights[ci], c.num_lookups,
                          c.args_width, bg, out, total_dev);
    return;
  }
  stage2_build_dyn(ctx, lk, bg, out, total_dev, &c.stage2_jit);
}
void stage2_circuit_async(Ctx& ctx, const HSystem& sys, const HWitness& wit, size_t ci, E2 beta, E2 gamma, u64* out, E2* total_dev) {
  DBuf<ChallengeBG> bg = challenge_bg_upload(ctx, beta, gamma);
  stage2_circuit_dyn(ctx, sys, wit, ci, bg.p, out, total_dev);
}

// claims, length-prefixed (src/prover.rs:369-373), absorbed into `ch`. Large claim sets are hashed on the device: the
// transcript since the last sample is `ch.input || words` and the next operation is a sample, so the digest is all the
// challenger needs (prove() below does the same with the stage-1 commitment patched in on the device).
void observe_claims(Ctx& ctx, Challenger& ch, HWitness& wit) {
  const size_t n_claims = wit.claim_offsets.size() - 1;
  const size_t claim_elems = wit.claim_data.size();
  const size_t claim_words = 1 + n_claims + claim_elems;
  if (claim_words <= 8192) {
    ch.observe((u64)n_claims);
    for (size_t i = 0; i < n_claims; i++) {
      size_t a = wit.claim_offsets[i], b = wit.claim_offsets[i + 1];
      ch.observe((u64)(b - a));
      for (size_t k = a; k < b; k++) ch.observe(wit.claim_data[k]);
    }
    return;
  }
  DBuf<uint8_t> d_prefix(ctx, ch.input.size());
  ctx.h2d(d_prefix.p, ch.input.data(), ch.input.size());
  DBuf<u64> d_words(ctx, claim_words);
  claims_transcript_words(ctx, wit.d_claim_data.p, wit.d_claim_offsets.p, n_claims, claim_elems, d_words.p);
  Digest d = blake3_device(ctx, d_prefix.p, ch.input.size(), d_words.p, claim_words);
  ch.flush_with(d);
}

// ------------------------------------------------------------------ prove
std::vector<uint8_t> prove(HSystem& sys, HWitness& wit, StageMs* times) {
  Ctx& ctx = *sys.ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  const Params& prm = sys.params;
  const unsigned lb = (unsigned)prm.log_blowup;
  const size_t C = sys.circuits.size();
  if (wit.sys != &sys || wit.heights.size() != C) throw std::runtime_error("witness does not belong to this system");
  if (wit.has_remote) throw std::runtime_error("this witness lacks traces that another rank computes: use ms_prove_sharded");
  double t_begin = now_ms(), t0;
  g_probes.start();
  auto lap = [&](int slot) {
    if (times) {
      ctx.sync();
      times->v[slot] = now_ms() - t0;
    }
  };

  RoctxRange whole("stark/prove");
  HostUpload up(wit, ctx);  // host-resident witness: uploads start now and run beside the transcript set-up
  up.row_groups = true;
  up.start();
  g_probes.mark("uploads issued");
  Challenger ch(sys.seed);
  // src/system.rs:211-222
  ch.observe((u64)C);
  for (auto& c : sys.circuits) {
    ch.observe((u64)c.constraint_count);
    ch.observe((u64)c.max_constraint_degree);
    ch.observe((u64)c.pre_height);
    ch.observe((u64)c.pre_width);
    ch.observe((u64)c.main_width);
    ch.observe((u64)c.stage2_width);
  }
  std::vector<uint8_t> active(C);
  std::vector<size_t> aidx;
  std::vector<int> apos(C, -1);
  for (size_t i = 0; i < C; i++) {
    active[i] = wit.heights[i] > 0;
    ch.observe(active[i] ? 1 : 0);
    if (active[i]) {
      apos[i] = (int)aidx.size();
      aidx.push_back(i);
    }
  }
  if (aidx.empty()) throw std::runtime_error("cannot prove with every circuit deactivated (all traces empty)");
  const size_t NA = aidx.size();
  std::vector<unsigned> log_degrees;
  for (size_t ci : aidx) log_degrees.push_back(log2_strict(wit.heights[ci]));
  // Short circuits next to long ones (the byte table beside 2^20 additions): a short circuit's kernels are a handful of
  // workgroups each and cost launch latency, not work, so within every phase they are queued on the context's side stream
  // AFTER the long circuits' launches and run beside those; the streams join again before the phase's commitment.
  std::vector<char> on_side(NA, 0);
  std::vector<size_t> order;  // positions: long circuits first
  ctx.side_config();
  {
    bool any_long = false, any_short = false;
    for (size_t pos = 0; pos < NA; pos++) (log_degrees[pos] <= ctx.side_max_log ? any_short : any_long) = true;
    const bool use_side = ctx.side_enabled && any_long && any_short;
    for (size_t pos = 0; pos < NA; pos++) on_side[pos] = use_side && log_degrees[pos] <= ctx.side_max_log;
    for (int pass = 0; pass < 2; pass++)
      for (size_t pos = 0; pos < NA; pos++)
        if ((int)on_side[pos] == pass) order.push_back(pos);
  }
  const bool use_side = order.size() && on_side[order.back()];
  struct SideSession {  // whatever happens, the streams are joined when prove() is left
    Ctx& c;
    ~SideSession() { c.side_join(); }
  } side_session{ctx};
  auto fork_side = [&]() {
    if (use_side && !ctx.side_forked) ctx.side_fork();
  };

  // claims, length-prefixed (src/prover.rs:369-373). Large claim sets are hashed on the device: the transcript since the
  // last sample is `ch.input || words`, and the next operation is a sample, so the digest is all the challenger needs.
  // The stage-1 commitment is part of that prefix; it is patched in on the device, so the commitment and the digest come
  // back in ONE synchronisation. Only the BLAKE3 chunks that hold prefix bytes depend on the commitment: the chaining
  // values of all the others - 41 000 of them at the bench size, latency-bound work - are computed on the side stream
  // while the stage-1 tree is hashed, and what is left behind the commitment is chunk 0 and the tree above the chunks.
  const size_t n_claims = wit.claim_offsets.size() - 1;
  const size_t claim_elems = wit.claim_data.size();
  const size_t claim_words = 1 + n_claims + claim_elems;
  const bool device_claims = claim_words > 8192;
  const bool early_claims = device_claims && ctx.side_enabled;
  size_t max_lde_log = 0;
  for (unsigned ld : log_degrees) max_lde_log = std::max<size_t>(max_lde_log, ld + lb);
  const size_t ncap = size_t(1) << std::min<size_t>((size_t)prm.cap_height, max_lde_log);
  std::vector<Digest> s1_cap;
  size_t cap_off = 0, prefix_chunks = 0, nchunks = 0;
  DBuf<u64> d_words;
  DBuf<Digest> d_cvs;
  // The outer transcript on the device (outer.hip): with the claims digest already computed there, beta/gamma, alpha and
  // zeta are sampled in the stream and every launch up to the opened-value sums is queued without a host round trip; the
  // host challenger replays the steps when the commitments arrive with the opened values, and is the authority.
  const bool dev_outer = device_claims && outer_fits(ncap, NA) && !getenv("MSAMD_HOST_TRANSCRIPT");
  struct OuterDev {
    DBuf<Digest> digest;
    DBuf<u32> state;  // 12 words behind gamma, 8 words behind alpha
    DBuf<E2> accs, alpha, points;
    DBuf<u32> lds;
    DBuf<uint8_t> circuits;
    std::vector<DBuf<uint8_t>> qdyn;  // per active circuit: QDyn, then the reversed alpha powers
    Digest h_digest;
    E2 h_bg[2], h_alpha;
    std::vector<E2> h_points;
    std::vector<unsigned> uniq_ld;  // the distinct trace heights: points[1 + k] = zeta * g(2^uniq_ld[k])
  } od;
  DBuf<ChallengeBG> d_bg;
  if (device_claims) {
    s1_cap.assign(ncap, Digest());
    if (sys.has_pre) ch.observe_cap(sys.pre_commit);
    cap_off = ch.input.size();
    ch.observe_cap(s1_cap);  // placeholder bytes: the real digests are copied over them on the device
    for (unsigned ld : log_degrees) ch.observe(ld);
    nchunks = blake3_num_chunks(ch.input.size(), claim_words);
    prefix_chunks = std::min(nchunks, (ch.input.size() + 1023) / 1024);
  }
  // chunk 0 alone depends on the commitment: the rest of the tree is hashed early too (hash.hip: blake3_late_chunk0_*)
  const bool late_chunk0 = device_claims && prefix_chunks == 1 && nchunks >= 2 && !getenv("MSAMD_OLD_CLAIMS_TREE");
  LateChunk0 late0;
  DBuf<uint8_t> d_prefix;
  auto claims_chunks = [&]() {  // everything of the claims digest that does not depend on the stage-1 commitment
    up.wait_claims();
    d_words = DBuf<u64>(ctx, claim_words);
    claims_transcript_words(ctx, wit.d_claim_data.p, wit.d_claim_offsets.p, n_claims, claim_elems, d_words.p);
    d_cvs = DBuf<Digest>(ctx, nchunks);
    blake3_chunk_cvs(ctx, nullptr, ch.input.size(), d_words.p, claim_words, prefix_chunks, nchunks, d_cvs.p);
    d_prefix = DBuf<uint8_t>(ctx, ch.input.size());
    ctx.h2d(d_prefix.p, ch.input.data(), ch.input.size());  // placeholder bytes where the commitment goes
    if (late_chunk0) blake3_late_chunk0_prepare(ctx, d_cvs.p, nchunks, late0);
  };

  // ---- stage 1 commit (src/prover.rs:336-351)
  t0 = now_ms();
  RoctxRange phase("stark/stage1_commit");
  PcsData s1;
  {
    std::vector<DMat> ldes(NA);
    up.wait_traces();
    fork_side();
    for (size_t pos : order) {
      const size_t ci = aidx[pos];
      SideScope sc(ctx, on_side[pos]);
      ldes[pos] = lde_of_host_matrix(ctx, wit.traces[ci].p, wit.heights[ci], sys.circuits[ci].main_width, lb,
                                     !on_side[pos] && ci < wit.early_evals.size() ? &wit.early_evals[ci] : nullptr);
    }
    ctx.side_join();
    if (early_claims) {
      // beside the leaf hashing and the tree levels of the commitment (integer-ALU work), not beside the transposes
      ctx.side_fork();
      SideScope sc(ctx, true);
      claims_chunks();
    }
    commit_matrices(ctx, std::move(ldes), (unsigned)prm.cap_height, s1);
  }
  g_probes.mark("stage 1 queued");
  up.wait_claims();
  if (!device_claims) {
    s1_cap = merkle_cap(ctx, s1.tree);
    lap(0);
    if (sys.has_pre) ch.observe_cap(sys.pre_commit);
    ch.observe_cap(s1_cap);
    for (unsigned ld : log_degrees) ch.observe(ld);
    ch.observe((u64)n_claims);
    for (size_t i = 0; i < n_claims; i++) {
      size_t a = wit.claim_offsets[i], b = wit.claim_offsets[i + 1];
      ch.observe((u64)(b - a));
      for (size_t k = a; k < b; k++) ch.observe(wit.claim_data[k]);
    }
  } else {
    lap(0);
    const size_t cl = s1.tree.cap_layer();
    if (s1.tree.layer_len[cl] != ncap) throw std::runtime_error("stage-1 cap size differs from the transcript's placeholder");
    const Digest* d_cap = s1.tree.base() + s1.tree.layer_off[cl];
    if (!early_claims) claims_chunks();
    ctx.side_join();
    od.digest = DBuf<Digest>(ctx, 1);
    if (dev_outer) {
      od.state = DBuf<u32>(ctx, 28);  // 12 words behind gamma, 8 behind alpha, 8 behind zeta
      d_bg = DBuf<ChallengeBG>(ctx, 1);
    }
    if (late_chunk0) {
      blake3_late_chunk0_finish(ctx, d_prefix.p, ch.input.size(), d_words.p, claim_words, d_cap, cap_off, ncap, late0, nchunks, od.digest.p,
                                dev_outer ? d_bg.p : nullptr, dev_outer ? od.state.p : nullptr);
    } else {
      HIP_CHECK(hipMemcpyAsync(d_prefix.p + cap_off, d_cap, ncap * sizeof(Digest), hipMemcpyDeviceToDevice, ctx.stream));
      blake3_chunk_cvs(ctx, d_prefix.p, ch.input.size(), d_words.p, claim_words, 0, prefix_chunks, d_cvs.p);
      blake3_from_cvs_async(ctx, d_cvs.p, nchunks, od.digest.p);
    }
    ctx.d2h_queue(s1_cap.data(), d_cap, ncap * sizeof(Digest));
    if (dev_outer) {
      if (!late_chunk0) outer_beta_gamma(ctx, od.digest.p, d_bg.p, od.state.p);
      ctx.d2h_queue(&od.h_digest, od.digest.p, sizeof(Digest));
      ctx.d2h_queue(od.h_bg, d_bg.p, 2 * sizeof(E2));  // beta, gamma lead the block
      g_probes.mark("claims digest + beta/gamma queued");
    } else {
      Digest d;
      ctx.d2h(&d, od.digest.p, sizeof(Digest));  // synchronises: s1_cap has arrived too
      g_probes.mark("sync 1 (cap + claims digest)");
      ch.flush_with(d);
    }
    d_words.reset();
    d_cvs.reset();
    late0.reset();
    d_prefix.reset();
  }
  E2 beta = e2(0), gamma = e2(0), alpha = e2(0), zeta = e2(0);
  auto tx_beta_gamma = [&]() {
    beta = ch.sample_ext();
    ch.observe_ext(beta);
    gamma = ch.sample_ext();
    ch.observe_ext(gamma);
  };
  if (!dev_outer) {
    tx_beta_gamma();
    d_bg = challenge_bg_upload(ctx, beta, gamma);
  }
  // initial accumulator from the claims (src/prover.rs:382-387). Nothing below needs the accumulators on the
  // host until they are observed after the stage-2 commitment, so the claims sum and every circuit's contribution
  // stay in device memory (d_tot[0] = claims, d_tot[1 + pos] = circuit) and come back with that commitment.
  DBuf<E2> d_tot(ctx, NA + 1);
  std::vector<E2> h_tot(NA + 1);
  // nothing reads the claims' sum before the accumulators are formed behind the stage-2 commitment: with a side stream it
  // runs there, beside the long circuits' stage-2 kernels (the streams join in front of that commitment) and BEHIND the short
  // circuits' stage 2, which the side stream also carries: the sum is a quarter of a millisecond of few, long-running waves
  // (one per SIMD); in front of the short circuits it held their launches back until the long circuit's stage-2 trace was
  // being written, and beside the long circuit's terms it cost that kernel 20 us (226 -> 206). MSAMD_CLAIMS_ACC_FIRST=1: the
  // earlier order
  const bool claims_on_device = n_claims > 256 || dev_outer;
  const bool claims_beside = claims_on_device && ctx.side_enabled && !getenv("MSAMD_CLAIMS_ACC_MAIN");
  const bool claims_last = claims_beside && !getenv("MSAMD_CLAIMS_ACC_FIRST");
  auto claims_sum_launch = [&]() {
    if (claims_beside && !ctx.side_forked) ctx.side_fork();
    SideScope sc(ctx, claims_beside);
    claims_accumulator_dyn(ctx, wit.d_claim_data.p, wit.d_claim_offsets.p, n_claims, d_bg.p, d_tot.p);
  };
  if (claims_on_device) {
    if (!claims_last) claims_sum_launch();
  } else {
    E2 acc0 = e2(0);
    for (size_t i = 0; i < n_claims; i++) {
      E2 f = e2(0);
      for (size_t k = wit.claim_offsets[i + 1]; k-- > wit.claim_offsets[i];) f = e2_add(e2_mul(f, gamma), e2(wit.claim_data[k]));
      acc0 = e2_add(acc0, e2_inv(e2_add(beta, f)));
    }
    ctx.h2d(d_tot.p, &acc0, sizeof(E2));
  }

  // ---- lookup construction (src/prover.rs:391-409) + stage 2 commit (:413-421)
  t0 = now_ms();
  phase.next("stark/lookup_construction");
  up.lookup_values();
  std::vector<DBuf<u64>> s2_evals(NA);
  fork_side();  // behind the claims accumulator and the lookup values: the side stream sees d_tot and the witness complete
  for (size_t pos : order) {
    size_t ci = aidx[pos];
    const HCircuit& c = sys.circuits[ci];
    size_t n = wit.heights[ci];
    SideScope sc(ctx, on_side[pos]);
    s2_evals[pos] = DBuf<u64>(ctx, n * c.stage2_width);
    stage2_circuit_dyn(ctx, sys, wit, ci, d_bg.p, s2_evals[pos].p, d_tot.p + 1 + pos);
  }
  if (claims_last) claims_sum_launch();
  lap(1);
  t0 = now_ms();
  phase.next("stark/stage2_commit");
  PcsData s2;
  {
    std::vector<DMat> ldes(NA);
    fork_side();
    for (size_t pos : order) {
      const HCircuit& c = sys.circuits[aidx[pos]];
      size_t n = wit.heights[aidx[pos]];
      SideScope sc(ctx, on_side[pos]);
      DMat lde;
      lde.h = n << lb;
      lde.w = c.stage2_width;
      lde.buf = DBuf<u64>(ctx, lde.h * lde.w);
      coset_lde(ctx, s2_evals[pos].p, lde.d(), log_degrees[pos], lb, lde.w);
      s2_evals[pos].reset();
      ldes[pos] = std::move(lde);
    }
    ctx.side_join();
    commit_matrices(ctx, std::move(ldes), (unsigned)prm.cap_height, s2);
  }
  ctx.d2h_queue(h_tot.data(), d_tot.p, (NA + 1) * sizeof(E2));
  g_probes.mark("stage 2 queued");
  auto dev_cap = [&](const DTree& t) -> const Digest* {
    const size_t cl = t.cap_layer();
    if (t.layer_len[cl] != ncap) throw std::runtime_error("a commitment's cap size differs from the stage-1 cap");
    return t.base() + t.layer_off[cl];
  };
  std::vector<Digest> s2_cap, q_cap;
  E2 acc_initial = e2(0);
  std::vector<E2> accs;
  auto tx_alpha = [&]() {  // src/prover.rs:382-433
    acc_initial = h_tot[0];
    accs.clear();
    E2 acc = acc_initial;
    for (size_t pos = 0; pos < NA; pos++) {
      acc = e2_add(acc, h_tot[1 + pos]);
      accs.push_back(acc);
    }
    ch.observe_cap(s2_cap);
    for (auto& a : accs) ch.observe_ext(a);
    alpha = ch.sample_ext();
  };
  if (dev_outer) {
    s2_cap.assign(ncap, Digest());
    accs.assign(NA, e2(0));
    const Digest* d_cap = dev_cap(s2.tree);
    ctx.d2h_queue(s2_cap.data(), d_cap, ncap * sizeof(Digest));
    od.accs = DBuf<E2>(ctx, NA + 1);
    od.alpha = DBuf<E2>(ctx, 1);
    od.qdyn.resize(NA);
    std::vector<OuterTarget> targets(NA);
    for (size_t pos = 0; pos < NA; pos++) {
      const size_t k = sys.circuits[aidx[pos]].prog.constraint_count;
      od.qdyn[pos] = DBuf<uint8_t>(ctx, sizeof(QDyn) + std::max<size_t>(k, 1) * sizeof(E2));
      targets[pos].dyn = reinterpret_cast<QDyn*>(od.qdyn[pos].p);
      targets[pos].alpha_rev = reinterpret_cast<E2*>(od.qdyn[pos].p + sizeof(QDyn));
      targets[pos].log_n = log_degrees[pos];
      targets[pos].k = k;
    }
    outer_alpha(ctx, od.state.p, d_cap, ncap, d_tot.p, NA, d_bg.p, targets, od.accs.p, od.alpha.p, od.state.p + 12, od.circuits);
    ctx.d2h_queue(&od.h_alpha, od.alpha.p, sizeof(E2));
    lap(2);
  } else {
    s2_cap = merkle_cap(ctx, s2.tree);  // synchronises: h_tot is complete as well
    g_probes.mark("sync 2 (stage-2 cap)");
    lap(2);
    tx_alpha();
  }

  // ---- quotient (src/prover.rs:437-528)
  t0 = now_ms();
  phase.next("stark/quotient");
  PcsData qd;
  {
    std::vector<DMat> qldes(NA);
    fork_side();
    for (size_t pos : order) {
      size_t ci = aidx[pos];
      const HCircuit& c = sys.circuits[ci];
      SideScope sc(ctx, on_side[pos]);
      unsigned log_n = log_degrees[pos], log_q = log2_strict(c.quotient_degree());
      size_t n = size_t(1) << log_n, nq = n << log_q;
      QuotientArgs qa;
      if (sys.has_pre && sys.pre_indices[ci] >= 0) {
        const DMat& pm = sys.pre_data.ldes[sys.pre_indices[ci]];
        qa.pre = pm.d();
        qa.pre_h = pm.h;
      }
      qa.s1 = s1.ldes[pos].d();
      qa.s1_h = s1.ldes[pos].h;
      qa.s2 = s2.ldes[pos].d();
      qa.s2_h = s2.ldes[pos].h;
      qa.log_n = log_n;
      qa.log_q = log_q;
      if (dev_outer) {
        qa.dyn = reinterpret_cast<const QDyn*>(od.qdyn[pos].p);
        qa.alpha_rev = reinterpret_cast<const E2*>(od.qdyn[pos].p + sizeof(QDyn));
      } else {
        const E2 four[4] = {beta, gamma, pos ? accs[pos - 1] : acc_initial, accs[pos]};
        for (int k = 0; k < 4; k++) {
          qa.publics[2 * k] = four[k].c0;
          qa.publics[2 * k + 1] = four[k].c1;
        }
        qa.alpha = alpha;
      }
      DBuf<u64> qv(ctx, nq * 2);
      quotient_eval(ctx, c.prog, qa, qv.p);
      DMat lde;
      lde.h = n << lb;
      lde.w = 2 << log_q;
      lde.buf = DBuf<u64>(ctx, lde.h * lde.w);
      quotient_lde(ctx, qv.p, lde.d(), log_n, log_q, lb, 2);
      qldes[pos] = std::move(lde);
    }
    ctx.side_join();
    commit_matrices(ctx, std::move(qldes), (unsigned)prm.cap_height, qd);
  }
  g_probes.mark("quotient queued");
  auto tx_zeta = [&]() {
    ch.observe_cap(q_cap);
    zeta = ch.sample_ext();
  };
  // zeta and zeta * g per trace height: values in host mode, placeholders for device values otherwise (pcs_open swaps them)
  E2 pt_zeta = e2(0);
  std::vector<E2> pt_next(NA);
  SymbolicPoints sym;
  if (dev_outer) {
    q_cap.assign(ncap, Digest());
    const Digest* d_cap = dev_cap(qd.tree);
    ctx.d2h_queue(q_cap.data(), d_cap, ncap * sizeof(Digest));
    std::vector<size_t> id_of(NA);
    for (size_t pos = 0; pos < NA; pos++) {
      size_t k = 0;
      while (k < od.uniq_ld.size() && od.uniq_ld[k] != log_degrees[pos]) k++;
      if (k == od.uniq_ld.size()) od.uniq_ld.push_back(log_degrees[pos]);
      id_of[pos] = 1 + k;
    }
    const size_t n_ld = od.uniq_ld.size();
    od.lds = DBuf<u32>(ctx, n_ld);
    ctx.h2d(od.lds.p, od.uniq_ld.data(), n_ld * sizeof(u32));
    od.points = DBuf<E2>(ctx, 1 + n_ld);
    od.h_points.assign(1 + n_ld, e2(0));
    outer_zeta(ctx, od.state.p + 12, d_cap, ncap, od.lds.p, n_ld, od.points.p, od.state.p + 20);
    ctx.d2h_queue(od.h_points.data(), od.points.p, (1 + n_ld) * sizeof(E2));
    pt_zeta = sym_point(0);
    for (size_t pos = 0; pos < NA; pos++) pt_next[pos] = sym_point(id_of[pos]);
    sym.d_points = od.points.p;
    sym.d_state = od.state.p + 20;
    sym.n = 1 + n_ld;
    sym.next_log.assign(1 + n_ld, -1);
    for (size_t k = 0; k < n_ld; k++) sym.next_log[1 + k] = (int)od.uniq_ld[k];
    lap(3);
    t0 = now_ms();
    phase.next("stark/fri_open");
  } else {
    q_cap = merkle_cap(ctx, qd.tree);
    g_probes.mark("sync 3 (quotient cap)");
    lap(3);
    // ---- opening (src/prover.rs:538-581)
    t0 = now_ms();
    phase.next("stark/fri_open");
    tx_zeta();
    pt_zeta = zeta;
    for (size_t pos = 0; pos < NA; pos++) pt_next[pos] = e2_mul_base(zeta, gl_two_adic_generator(log_degrees[pos]));
  }
  std::vector<OpenRound> rounds(3);
  rounds[0].data = &s1;
  rounds[1].data = &s2;
  rounds[2].data = &qd;
  for (size_t pos = 0; pos < NA; pos++) {
    rounds[0].points.push_back({pt_zeta, pt_next[pos]});
    rounds[1].points.push_back({pt_zeta, pt_next[pos]});
    rounds[2].points.push_back({pt_zeta});
  }
  if (sys.has_pre) {
    OpenRound r0;
    r0.data = &sys.pre_data;
    for (size_t ci = 0; ci < C; ci++) {
      if (sys.pre_indices[ci] < 0) continue;
      if (apos[ci] >= 0) {
        r0.points.push_back({pt_zeta, pt_next[apos[ci]]});
      } else {
        r0.points.push_back({});
      }
    }
    rounds.push_back(std::move(r0));
  }
  // ---- Proof bytes, field order of src/prover.rs:213-238: the fields in front of the opening proof are known by now,
  // so the FriProof is serialised straight behind them (no second copy of its ~1 MB)
  PW w;
  {
    size_t vals = 0;
    for (size_t pos = 0; pos < NA; pos++) {
      const HCircuit& c = sys.circuits[aidx[pos]];
      vals += 2 * (c.main_width + c.stage2_width + c.pre_width + 2) + 2 * c.quotient_degree() + 16;
    }
    w.b.reserve(16 * vals + 64 * (NA + 8) + 32 * (s1_cap.size() + s2_cap.size() + q_cap.size()) + 8192);
  }
  auto write_header = [&](PW& h) {
    h.u64_(C);
    for (auto a : active) h.u8(a);
    h.cap(s1_cap);
    h.cap(s2_cap);
    h.cap(q_cap);
    h.u64_(accs.size());
    for (auto& a : accs) h.ext(a);
    h.u64_(log_degrees.size());
    for (unsigned ld : log_degrees) h.u8((uint8_t)ld);
  };
  write_header(w);  // device transcript: same length, contents still on their way - rewritten in `resolve`
  const size_t header_len = w.b.size();
  sym.resolve = [&](std::vector<E2>& values) {
    // everything queued since the stage-1 commitment has arrived: replay the transcript on the host and compare
    ch.flush_with(od.h_digest);
    tx_beta_gamma();
    tx_alpha();
    tx_zeta();
    bool same = e2_same(beta, od.h_bg[0]) && e2_same(gamma, od.h_bg[1]) && e2_same(alpha, od.h_alpha) && e2_same(zeta, od.h_points[0]);
    values[0] = zeta;
    for (size_t k = 0; k < od.uniq_ld.size(); k++) {
      values[1 + k] = e2_mul_base(zeta, gl_two_adic_generator(od.uniq_ld[k]));
      same = same && e2_same(values[1 + k], od.h_points[1 + k]);
    }
    if (!same) throw std::runtime_error("the device transcript's challenges differ from the host challenger's");
    PW h;
    write_header(h);
    if (h.b.size() != header_len) throw std::runtime_error("proof header changed length");
    memcpy(w.b.data(), h.b.data(), header_len);
    g_probes.mark("transcript replayed");
  };
  std::vector<OpenedRound> opened;
  pcs_open(sys, rounds, ch, opened, w, dev_outer ? &sym : nullptr);
  lap(4);
  write_round(w, opened[2]);
  w.u8(sys.has_pre ? 1 : 0);
  if (sys.has_pre) write_round(w, opened[3]);
  write_round(w, opened[0]);
  write_round(w, opened[1]);
  ctx.prof_collect();
  if (times) times->v[5] = now_ms() - t_begin;
  g_probes.mark("proof bytes complete");
  g_probes.print();
  if (getenv("MSAMD_TRACE_HOST")) fprintf(stderr, "[msamd] prove(): %.1f us in all\n", 1e3 * (now_ms() - t_begin));
  return std::move(w.b);
}

#include "prover_sharded.inc"

// Pcs::open on its own (src/prover.rs:580; examples/pcs_example.rs:88-95): the same code path as inside prove()
void pcs_open_standalone(Ctx& ctx, const Params& prm, const std::vector<PcsData*>& data, const std::vector<std::vector<std::vector<E2>>>& points,
                         Challenger& ch, std::vector<E2>& opened_flat, std::vector<uint8_t>& fri_bytes) {
  HIP_CHECK(hipSetDevice(ctx.device));
  HSystem sys;  // carries the context and the parameters only
  sys.ctx = &ctx;
  sys.params = prm;
  if (prm.max_log_arity < 1 || prm.max_log_arity > FRI_MAX_LOG_ARITY) throw std::runtime_error("max_log_arity must be 1 .. 6 (a FRI row is hashed as one BLAKE3 chunk)");
  if (prm.log_blowup < 1 || prm.log_blowup > 8) throw std::runtime_error("log_blowup out of range");
  if (prm.commit_pow_bits > 40 || prm.query_pow_bits > 40) throw std::runtime_error("proof-of-work bits out of range");
  std::vector<OpenRound> rounds;
  for (size_t r = 0; r < data.size(); r++) {
    if (points[r].size() != data[r]->ldes.size()) throw std::runtime_error("pcs_open: one list of points per matrix expected");
    for (auto& m : data[r]->ldes)
      if (log2_strict(m.h) < prm.log_blowup) throw std::runtime_error("pcs_open: a committed matrix is shorter than the blowup");
    rounds.push_back(OpenRound{data[r], points[r]});
  }
  std::vector<OpenedRound> opened;
  PW fri;
  pcs_open(sys, rounds, ch, opened, fri);
  opened_flat.clear();
  for (auto& orr : opened)
    for (auto& m : orr)
      for (auto& pt : m) opened_flat.insert(opened_flat.end(), pt.begin(), pt.end());
  fri_bytes = std::move(fri.b);
  ctx.sync();
}

}  // namespace msamd
