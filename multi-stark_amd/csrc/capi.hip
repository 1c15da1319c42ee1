// extern "C" boundary (include/mstark.h). No C++ exception crosses it.
#include <string>

#include "../../include/mstark.h"
#include "host.h"

using namespace msamd;

// Handles keep what they depend on alive: a system holds a reference on its context, a witness on its system, an
// mmcs on its context. Destroying a handle drops the owner's reference; the object goes away when the last dependent
// is gone, so handles may be destroyed in any order (a garbage collector gives none). A context is single-threaded by
// contract, so the counts are plain integers.
struct ms_ctx {
  Ctx ctx;
  int refs = 1;
  explicit ms_ctx(int dev) : ctx(dev) {}
};
static void ctx_unref(ms_ctx* c) {
  if (c && --c->refs == 0) delete c;
}
struct ms_system {
  ms_ctx* owner = nullptr;
  std::unique_ptr<HSystem> sys;
  int refs = 1;
};
static void system_unref(ms_system* s) {
  if (s && --s->refs == 0) {
    ms_ctx* c = s->owner;
    s->sys.reset();  // device buffers go back to the context's pool before the context may be deleted
    delete s;
    ctx_unref(c);
  }
}
struct ms_witness {
  ms_system* owner = nullptr;
  std::unique_ptr<HWitness> w;
};
struct ms_mmcs {
  ms_ctx* owner = nullptr;
  Ctx* ctx = nullptr;
  PcsData data;
  // a view of prover data owned by a system (the preprocessed commitment, ms_system_preprocessed_mmcs): `borrowed` points at
  // it and `lender` keeps the system alive
  PcsData* borrowed = nullptr;
  ms_system* lender = nullptr;
  PcsData& pd() { return borrowed ? *borrowed : data; }
};
// a device-resident matrix handed between Level-2 calls: stage-2 evaluations (kind 0: n x w, rows in the order the
// inverse transform wants) or a committed-to-be LDE (kind 1: (n << log_blowup) x w)
struct ms_trace {
  ms_ctx* owner = nullptr;
  Ctx* ctx = nullptr;
  DMat m;
  unsigned log_n = 0;
  int kind = 0;
};

static thread_local std::string g_err;

#define MS_TRY try {
#define MS_CATCH                    \
  }                                 \
  catch (const std::exception& e) { \
    g_err = e.what();               \
    (void)hipGetLastError();        \
    msamd::abandon_pending();       \
    return MS_ERR;                  \
  }                                 \
  catch (...) {                     \
    g_err = "unknown error";        \
    msamd::abandon_pending();       \
    return MS_ERR;                  \
  }

namespace {
// host row-major -> device column-major (optionally bit-reversed rows)
DBuf<u64> upload_colmajor(Ctx& ctx, const u64* host, size_t h, size_t w, bool bitrev_rows) {
  for (size_t i = 0; i < h * w; i++)
    if (host[i] >= GL_P) throw std::runtime_error("non-canonical field element in input");
  DBuf<u64> raw(ctx, h * w), col(ctx, h * w);
  ctx.h2d(raw.p, host, h * w * 8);
  transpose_in(ctx, raw.p, col.p, h, w, bitrev_rows);
  ctx.sync();
  return col;
}
void download_rowmajor(Ctx& ctx, const u64* col, size_t h, size_t w, bool bitrev_rows, u64* host) {
  DBuf<u64> row(ctx, h * w);
  transpose_out(ctx, col, row.p, h, w, bitrev_rows);
  ctx.d2h(host, row.p, h * w * 8);
}
// an extension-field input of a Level-2 call: both coordinates canonical
E2 canonical_e2(const uint64_t v[2], const char* what) {
  if (!v) throw std::runtime_error(std::string(what) + ": null");
  if (v[0] >= GL_P || v[1] >= GL_P) throw std::runtime_error(std::string("non-canonical ") + what);
  return e2(v[0], v[1]);
}
void check_pow2(size_t h) {
  if (h == 0 || (h & (h - 1))) throw std::runtime_error("height must be a power of two");
}
}  // namespace

// hooks for the other C-ABI translation unit (bb_prover.hip, include/mstark_bb.h): one error string, one context type
namespace msamd {
void set_last_error(const char* what) {
  g_err = what;
  (void)hipGetLastError();
  abandon_pending();
}
Ctx* ctx_of(ms_ctx* c) {
  if (!c) throw std::runtime_error("null context");
  return &c->ctx;
}
std::string last_error_text() { return g_err; }
void clear_last_error() { g_err.clear(); }
void ctx_retain(ms_ctx* c) { c->refs++; }
void ctx_release(ms_ctx* c) { ctx_unref(c); }
}  // namespace msamd

extern "C" {

const char* ms_last_error(void) { return g_err.c_str(); }

int32_t ms_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}
int32_t ms_ctx_create(int32_t device, ms_ctx** out) {
  try {
    *out = new ms_ctx(device);
    return MS_OK;
  } catch (const std::exception& e) {
    g_err = e.what();
    *out = nullptr;
    return MS_ERR_NO_DEVICE;
  }
}
void ms_ctx_destroy(ms_ctx* ctx) { ctx_unref(ctx); }
int32_t ms_ctx_sync(ms_ctx* ctx) {
  MS_TRY ctx->ctx.sync();
  return MS_OK;
  MS_CATCH
}
int32_t ms_ctx_sync_count(ms_ctx* ctx, uint64_t* out) {
  if (!ctx || !out) return MS_ERR;
  *out = ctx->ctx.host_syncs;
  return MS_OK;
}
int32_t ms_ctx_trim(ms_ctx* ctx) {
  MS_TRY ctx->ctx.trim();
  return MS_OK;
  MS_CATCH
}
int32_t ms_ctx_set_profile_mask(ms_ctx* ctx, uint32_t mask) {
  ctx->ctx.prof_mask = mask;
  return MS_OK;
}
int32_t ms_ctx_kernel_stats(ms_ctx* ctx, int32_t id, uint64_t* launches, double* ms, double* alg_bytes) {
  MS_TRY if (id < 0 || id >= K_COUNT) throw std::runtime_error("kernel id out of range");
  ctx->ctx.prof_collect();
  const KernelStat& s = ctx->ctx.stats[id];
  *launches = s.launches;
  *ms = s.ms;
  *alg_bytes = s.alg_bytes;
  return MS_OK;
  MS_CATCH
}
int32_t ms_ctx_kernel_units(ms_ctx* ctx, int32_t id, double* units) {
  MS_TRY if (id < 0 || id >= K_COUNT) throw std::runtime_error("kernel id out of range");
  ctx->ctx.prof_collect();
  *units = ctx->ctx.stats[id].units;
  return MS_OK;
  MS_CATCH
}
int32_t ms_ctx_reset_stats(ms_ctx* ctx) {
  MS_TRY ctx->ctx.prof_collect();
  for (auto& s : ctx->ctx.stats) s = KernelStat();
  return MS_OK;
  MS_CATCH
}
int32_t ms_ctx_debug_fail_alloc(ms_ctx* ctx, int32_t nth) {
  ctx->ctx.fail_alloc_countdown = nth > 0 ? nth : 0;
  return MS_OK;
}
int32_t ms_kernel_count(void) { return K_COUNT; }
const char* ms_kernel_name(int32_t id) { return kernel_name(id); }

int32_t ms_system_create(ms_ctx* ctx, const uint8_t* blob, size_t len, ms_system** out) {
  *out = nullptr;
  MS_TRY std::unique_ptr<ms_system> s(new ms_system());
  s->sys = system_from_blob(ctx->ctx, blob, len);
  s->owner = ctx;
  ctx->refs++;
  *out = s.release();
  return MS_OK;
  MS_CATCH
}
void ms_system_destroy(ms_system* sys) { system_unref(sys); }
int32_t ms_system_preprocessed_commit(const ms_system* sys, uint8_t* out, size_t cap, size_t* n_digests) {
  MS_TRY const HSystem& s = *sys->sys;
  *n_digests = s.has_pre ? s.pre_commit.size() : 0;
  if (*n_digests * 32 > cap) return MS_ERR_BUFFER;
  for (size_t i = 0; i < *n_digests; i++) memcpy(out + 32 * i, s.pre_commit[i].b, 32);
  return MS_OK;
  MS_CATCH
}
int32_t ms_system_circuit_info(const ms_system* sys, size_t ci, uint64_t out9[9]) {
  MS_TRY const HSystem& s = *sys->sys;
  if (ci >= s.circuits.size()) throw std::runtime_error("circuit index out of range");
  const HCircuit& c = s.circuits[ci];
  const uint64_t v[9] = {c.main_width,       c.pre_width,           c.pre_height,         c.num_lookups, c.stage2_width,
                         c.constraint_count, c.max_constraint_degree, c.quotient_degree(), c.args_width};
  memcpy(out9, v, sizeof(v));
  return MS_OK;
  MS_CATCH
}

int32_t ms_witness_create(ms_system* sys, const uint64_t* const* traces, const uint64_t* heights, const uint64_t* const* mult,
                          const uint64_t* const* args, size_t n_claims, const uint64_t* claim_offsets,
                          const uint64_t* claim_data, ms_witness** out) {
  *out = nullptr;
  MS_TRY std::unique_ptr<ms_witness> w(new ms_witness());
  w->w = witness_create(*sys->sys, traces, heights, mult, args, n_claims, claim_offsets, claim_data);
  w->owner = sys;
  sys->refs++;
  *out = w.release();
  return MS_OK;
  MS_CATCH
}
int32_t ms_witness_create_host(ms_system* sys, const uint64_t* const* traces, const uint64_t* heights, size_t n_claims,
                               const uint64_t* claim_offsets, const uint64_t* claim_data, int32_t* pinned, ms_witness** out) {
  *out = nullptr;
  MS_TRY std::unique_ptr<ms_witness> w(new ms_witness());
  w->w = witness_create_host(*sys->sys, traces, heights, n_claims, claim_offsets, claim_data);
  if (pinned) *pinned = w->w->pinned ? 1 : 0;
  w->owner = sys;
  sys->refs++;
  *out = w.release();
  return MS_OK;
  MS_CATCH
}
int32_t ms_claims_slice_range(ms_system* sys, const uint64_t* heights, size_t n_claims, const uint64_t* claim_offsets, int32_t rank,
                              int32_t world, uint64_t* first_elem, uint64_t* n_elems) {
  MS_TRY if (!sys || !heights || !claim_offsets || !first_elem || !n_elems) throw std::runtime_error("ms_claims_slice_range: null argument");
  if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("ms_claims_slice_range: rank out of range");
  if (claim_offsets[0] != 0) throw std::runtime_error("claim offsets must start at 0");
  std::vector<size_t> hs(heights, heights + sys->sys->circuits.size());
  for (size_t h : hs)
    if (h & (h - 1)) throw std::runtime_error("trace height must be a power of two");
  size_t first = 0, count = 0;
  claims_slice_range(*sys->sys, hs.data(), n_claims, claim_offsets, (size_t)rank, (size_t)world, first, count);
  *first_elem = first;
  *n_elems = count;
  return MS_OK;
  MS_CATCH
}
int32_t ms_witness_create_host_sliced(ms_system* sys, const uint64_t* const* traces, const uint64_t* heights, size_t n_claims,
                                      const uint64_t* claim_offsets, uint64_t data_first, uint64_t data_count,
                                      const uint64_t* data_slice, const uint64_t* head, size_t n_head, int32_t* pinned,
                                      ms_witness** out) {
  *out = nullptr;
  MS_TRY std::unique_ptr<ms_witness> w(new ms_witness());
  if (data_count == ~uint64_t(0)) throw std::runtime_error("ms_witness_create_host_sliced: data_count out of range");
  w->w = witness_create_host(*sys->sys, traces, heights, n_claims, claim_offsets, data_slice, (size_t)data_first, (size_t)data_count, head, n_head);
  if (pinned) *pinned = w->w->pinned ? 1 : 0;
  w->owner = sys;
  sys->refs++;
  *out = w.release();
  return MS_OK;
  MS_CATCH
}
int32_t ms_witness_prefetch(ms_witness* w, int32_t on) {
  MS_TRY HWitness& wit = *w->w;
  if (!wit.host_resident) throw std::runtime_error("ms_witness_prefetch: only a host-resident witness is uploaded per proof");
  Ctx& ctx = *wit.sys->ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  wit.prefetch = on != 0;
  if (!wit.prefetch) {
    HIP_CHECK(hipStreamSynchronize(ctx.copy_stream));
    HIP_CHECK(hipStreamSynchronize(ctx.claims_stream));
    for (auto& st : wit.stage) st.clear();
  }
  return MS_OK;
  MS_CATCH
}
int32_t ms_witness_u32_add_bench(ms_system* sys, size_t num_adds, uint32_t a0, uint32_t b0, ms_witness** out) {
  *out = nullptr;
  MS_TRY std::unique_ptr<ms_witness> w(new ms_witness());
  w->w = witness_u32_add_bench(*sys->sys, num_adds, a0, b0);
  w->owner = sys;
  sys->refs++;
  *out = w.release();
  return MS_OK;
  MS_CATCH
}
void ms_witness_destroy(ms_witness* w) {
  if (!w) return;
  ms_system* s = w->owner;
  w->w.reset();
  delete w;
  system_unref(s);
}

int32_t ms_prove(ms_system* sys, ms_witness* w, uint8_t* proof_out, size_t cap, size_t* proof_len, double* stage_ms) {
  MS_TRY StageMs st;
  std::vector<uint8_t> bytes = prove(*sys->sys, *w->w, stage_ms ? &st : nullptr);
  if (stage_ms) memcpy(stage_ms, st.v, sizeof(st.v));
  *proof_len = bytes.size();
  if (bytes.size() > cap) return MS_ERR_BUFFER;
  memcpy(proof_out, bytes.data(), bytes.size());
  return MS_OK;
  MS_CATCH
}

int32_t ms_prove_sharded(ms_system* sys, ms_witness* w, const ms_comm* comm, const int32_t* owners, uint8_t* proof_out, size_t cap,
                         size_t* proof_len, double* stage_ms) {
  MS_TRY StageMs st;
  if (!comm || !owners) throw std::runtime_error("ms_prove_sharded: null argument");
  // the host's table may be shorter than this library's (ms_comm.size): members beyond it are "not offered"
  if (comm->size < offsetof(ms_comm, all_gather) + sizeof(comm->all_gather))
    throw std::runtime_error("ms_prove_sharded: ms_comm.size is not set (sizeof(ms_comm) of the host's header) or too small");
  ms_comm table;
  memset(&table, 0, sizeof(table));
  memcpy(&table, comm, std::min<size_t>(comm->size, sizeof(table)));
  table.size = (uint32_t)sizeof(table);
  if (!table.all_to_all || !table.all_gather) throw std::runtime_error("ms_prove_sharded: incomplete ms_comm");
  if (table.rank < 0 || table.rank >= table.world) throw std::runtime_error("ms_prove_sharded: rank out of range");
  g_err.clear();  // (a failing transport callback leaves its reason here; prove_sharded quotes it)
  std::vector<uint8_t> bytes;
  try {
    bytes = prove_sharded(*sys->sys, *w->w, &table, owners, stage_ms ? &st : nullptr);
  } catch (const std::exception& e) {
    // this rank leaves the proof: its peers are in, or about to enter, an exchange it will never join
    if (table.abort && table.world > 1) table.abort(table.user, e.what());
    throw;
  }
  if (stage_ms) memcpy(stage_ms, st.v, sizeof(st.v));
  *proof_len = bytes.size();
  if (bytes.size() > cap) return MS_ERR_BUFFER;
  memcpy(proof_out, bytes.data(), bytes.size());
  return MS_OK;
  MS_CATCH
}

int32_t ms_ctx_comm_progress(ms_ctx* ctx, char* out, size_t cap, uint64_t* seq, int32_t* in_flight) {
  if (!ctx) return MS_ERR;
  std::lock_guard<std::mutex> lk(ctx->ctx.comm_mu);
  if (out && cap) {
    const size_t n = std::min(cap - 1, ctx->ctx.comm_what.size());
    memcpy(out, ctx->ctx.comm_what.data(), n);
    out[n] = 0;
  }
  if (seq) *seq = ctx->ctx.comm_seq;
  if (in_flight) *in_flight = ctx->ctx.comm_in_flight ? 1 : 0;
  return MS_OK;
}

int32_t ms_verify(ms_system* sys, size_t n_claims, const uint64_t* claim_offsets, const uint64_t* claim_data, const uint8_t* proof,
                  size_t proof_len, int32_t* verdict) {
  MS_TRY if (!verdict || !proof || !claim_offsets) throw std::runtime_error("ms_verify: null argument");
  if (claim_offsets[0] != 0) throw std::runtime_error("claim offsets must start at 0");
  *verdict = verify(*sys->sys, n_claims, claim_offsets, claim_data, proof, proof_len);
  return MS_OK;
  MS_CATCH
}

int32_t ms_dft_batch(ms_ctx* c, const uint64_t* in, size_t h, size_t w, int32_t inverse, uint64_t* out) {
  MS_TRY Ctx& ctx = c->ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  check_pow2(h);
  if (w == 0) return MS_OK;
  unsigned logn = log2_strict(h);
  DBuf<u64> col = upload_colmajor(ctx, in, h, w, true);  // bit-reversed rows feed the in-place DIT
  u64 scale = inverse ? gl_inv((u64)h % GL_P) : 1;
  ntt_dit(ctx, col.p, logn, w, inverse != 0, scale);
  download_rowmajor(ctx, col.p, h, w, false, out);
  return MS_OK;
  MS_CATCH
}

int32_t ms_coset_lde_batch(ms_ctx* c, const uint64_t* in, size_t h, size_t w, uint32_t log_blowup, uint64_t* out) {
  MS_TRY Ctx& ctx = c->ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  check_pow2(h);
  if (w == 0) return MS_OK;
  unsigned logn = log2_strict(h);
  if (logn > NTT_MAX_LOG || logn + log_blowup > TW_LOG) throw std::runtime_error("matrix too tall");
  DBuf<u64> col = upload_colmajor(ctx, in, h, w, true);
  DBuf<u64> lde(ctx, (h << log_blowup) * w);
  coset_lde(ctx, col.p, lde.p, logn, log_blowup, w);
  download_rowmajor(ctx, lde.p, h << log_blowup, w, false, out);
  return MS_OK;
  MS_CATCH
}

int32_t ms_quotient_lde(ms_ctx* c, const uint64_t* in, uint32_t log_n, uint32_t log_q, uint32_t log_blowup, size_t D,
                        uint64_t* out) {
  MS_TRY Ctx& ctx = c->ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  size_t nq = size_t(1) << (log_n + log_q);
  if (log_n + log_q > NTT_MAX_LOG || log_n + log_blowup > TW_LOG) throw std::runtime_error("matrix too tall");
  DBuf<u64> col = upload_colmajor(ctx, in, nq, D, true);  // storage (bit-reversed) order, as the quotient kernel writes
  size_t H = size_t(1) << (log_n + log_blowup), W = D << log_q;
  DBuf<u64> lde(ctx, H * W);
  quotient_lde(ctx, col.p, lde.p, log_n, log_q, log_blowup, D);
  download_rowmajor(ctx, lde.p, H, W, false, out);
  return MS_OK;
  MS_CATCH
}

int32_t ms_mmcs_commit(ms_ctx* c, size_t n, const uint64_t* const* mats, const uint64_t* heights, const uint64_t* widths,
                       uint32_t cap_height, uint8_t* cap_out, ms_mmcs** out) {
  *out = nullptr;
  MS_TRY Ctx& ctx = c->ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  std::unique_ptr<ms_mmcs> m(new ms_mmcs());
  m->ctx = &ctx;
  std::vector<DMat> ms;
  for (size_t i = 0; i < n; i++) {
    check_pow2(heights[i]);
    DMat dm;
    dm.h = heights[i];
    dm.w = widths[i];
    dm.buf = upload_colmajor(ctx, mats[i], dm.h, dm.w, false);
    ms.push_back(std::move(dm));
  }
  commit_matrices(ctx, std::move(ms), cap_height, m->data);
  std::vector<Digest> cap = merkle_cap(ctx, m->data.tree);
  for (size_t i = 0; i < cap.size(); i++) memcpy(cap_out + 32 * i, cap[i].b, 32);
  m->owner = c;
  c->refs++;
  *out = m.release();
  return MS_OK;
  MS_CATCH
}

// ---- Pcs::commit / Pcs::open / Pcs::verify on their own, with the challenger as a handle
int32_t ms_pcs_commit(ms_ctx* c, uint32_t log_blowup, uint32_t cap_height, size_t n, const uint64_t* const* evals, const uint64_t* heights,
                      const uint64_t* widths, uint8_t* cap_out, ms_mmcs** out) {
  *out = nullptr;
  MS_TRY Ctx& ctx = c->ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  if (log_blowup < 1 || log_blowup > 8) throw std::runtime_error("log_blowup out of range");
  std::unique_ptr<ms_mmcs> m(new ms_mmcs());
  m->ctx = &ctx;
  std::vector<DMat> ms;
  for (size_t i = 0; i < n; i++) {
    check_pow2(heights[i]);
    unsigned logn = log2_strict(heights[i]);
    if (logn > NTT_MAX_LOG || logn + log_blowup > TW_LOG) throw std::runtime_error("matrix too tall");
    DBuf<u64> col = upload_colmajor(ctx, evals[i], heights[i], widths[i], true);
    DMat dm;
    dm.h = (size_t)heights[i] << log_blowup;
    dm.w = widths[i];
    dm.buf = DBuf<u64>(ctx, dm.h * dm.w);
    if (dm.w) coset_lde(ctx, col.p, dm.buf.p, logn, log_blowup, dm.w);
    ctx.sync();
    ms.push_back(std::move(dm));
  }
  commit_matrices(ctx, std::move(ms), cap_height, m->data);
  std::vector<Digest> cap = merkle_cap(ctx, m->data.tree);
  for (size_t i = 0; i < cap.size(); i++) memcpy(cap_out + 32 * i, cap[i].b, 32);
  m->owner = c;
  c->refs++;
  *out = m.release();
  return MS_OK;
  MS_CATCH
}

static Params params_from7(const uint64_t* p7) {
  Params p;
  p.log_blowup = p7[0], p.cap_height = p7[1], p.log_final_poly_len = p7[2], p.max_log_arity = p7[3], p.num_queries = p7[4];
  p.commit_pow_bits = p7[5], p.query_pow_bits = p7[6];
  return p;
}
struct ms_challenger {
  Challenger ch;
  explicit ms_challenger(const std::vector<uint8_t>& seed) : ch(seed) {}
};
int32_t ms_challenger_create(const uint64_t params7[7], ms_challenger** out) {
  MS_TRY const char* tag = "multi-stark/v0";  // src/types.rs:118-130
  std::vector<uint8_t> seed(tag, tag + 14);
  for (int i = 0; i < 7; i++)
    for (int k = 0; k < 8; k++) seed.push_back((uint8_t)(params7[i] >> (8 * k)));
  *out = new ms_challenger(seed);
  return MS_OK;
  MS_CATCH
}
void ms_challenger_destroy(ms_challenger* ch) { delete ch; }
int32_t ms_challenger_observe(ms_challenger* ch, const uint64_t* elems, size_t n) {
  MS_TRY for (size_t i = 0; i < n; i++) {
    if (elems[i] >= GL_P) throw std::runtime_error("non-canonical field element");
    ch->ch.observe(elems[i]);
  }
  return MS_OK;
  MS_CATCH
}
int32_t ms_challenger_observe_digests(ms_challenger* ch, const uint8_t* digests, size_t n) {
  MS_TRY ch->ch.observe_bytes(digests, 32 * n);
  return MS_OK;
  MS_CATCH
}
int32_t ms_challenger_sample_ext(ms_challenger* ch, uint64_t out2[2]) {
  MS_TRY E2 e = ch->ch.sample_ext();
  out2[0] = e.c0, out2[1] = e.c1;
  return MS_OK;
  MS_CATCH
}
int32_t ms_challenger_sample_bits(ms_challenger* ch, uint32_t bits, uint64_t* out) {
  MS_TRY if (bits > 63) throw std::runtime_error("too many bits");
  *out = ch->ch.sample_bits(bits);
  return MS_OK;
  MS_CATCH
}
int32_t ms_pcs_open(ms_ctx* c, const uint64_t params7[7], size_t n_rounds, ms_mmcs* const* rounds, const uint64_t* n_points, const uint64_t* points,
                    ms_challenger* ch, uint64_t* opened_out, size_t opened_cap_words, uint8_t* fri_out, size_t fri_cap, size_t* fri_len) {
  MS_TRY Ctx& ctx = c->ctx;
  std::vector<PcsData*> data;
  std::vector<std::vector<std::vector<E2>>> pts(n_rounds);
  size_t mk = 0, pk = 0;
  for (size_t r = 0; r < n_rounds; r++) {
    if (rounds[r]->ctx != &ctx) throw std::runtime_error("commitment belongs to another context");
    data.push_back(&rounds[r]->pd());
    for (size_t m = 0; m < rounds[r]->pd().ldes.size(); m++) {
      std::vector<E2> pl;
      for (uint64_t k = 0; k < n_points[mk]; k++) {
        if (points[2 * pk] >= GL_P || points[2 * pk + 1] >= GL_P) throw std::runtime_error("non-canonical opening point");
        pl.push_back(E2{points[2 * pk], points[2 * pk + 1]});
        pk++;
      }
      mk++;
      pts[r].push_back(std::move(pl));
    }
  }
  std::vector<E2> opened;
  std::vector<uint8_t> fri;
  pcs_open_standalone(ctx, params_from7(params7), data, pts, ch->ch, opened, fri);
  *fri_len = fri.size();
  if (opened.size() * 2 > opened_cap_words || fri.size() > fri_cap) return MS_ERR_BUFFER;
  for (size_t i = 0; i < opened.size(); i++) opened_out[2 * i] = opened[i].c0, opened_out[2 * i + 1] = opened[i].c1;
  memcpy(fri_out, fri.data(), fri.size());
  return MS_OK;
  MS_CATCH
}
int32_t ms_pcs_verify(const uint64_t params7[7], size_t n_rounds, const uint8_t* const* caps, const uint64_t* cap_sizes, const uint64_t* n_mats,
                      const uint64_t* log_n, const uint64_t* widths, const uint64_t* n_points, const uint64_t* points, const uint64_t* opened,
                      const uint8_t* fri, size_t fri_len, ms_challenger* ch, int32_t* accepted) {
  MS_TRY std::vector<std::vector<Digest>> commits(n_rounds);
  std::vector<std::vector<unsigned>> ln(n_rounds);
  std::vector<std::vector<size_t>> ws(n_rounds);
  std::vector<std::vector<std::vector<E2>>> pts(n_rounds);
  size_t mk = 0, pk = 0, total = 0;
  *accepted = 0;
  for (size_t r = 0; r < n_rounds; r++) {
    commits[r].resize(cap_sizes[r]);
    for (size_t i = 0; i < cap_sizes[r]; i++) memcpy(commits[r][i].b, caps[r] + 32 * i, 32);
    for (uint64_t m = 0; m < n_mats[r]; m++) {
      if (log_n[mk] + params7[0] > GL_TWO_ADICITY) return MS_OK;  // rejected: the LDE domain would exceed the field's 2-adicity
      ln[r].push_back((unsigned)log_n[mk]);
      ws[r].push_back((size_t)widths[mk]);
      std::vector<E2> pl;
      for (uint64_t k = 0; k < n_points[mk]; k++) {
        if (points[2 * pk] >= GL_P || points[2 * pk + 1] >= GL_P) return MS_OK;
        pl.push_back(E2{points[2 * pk], points[2 * pk + 1]});
        pk++;
      }
      total += pl.size() * widths[mk];
      mk++;
      pts[r].push_back(std::move(pl));
    }
  }
  std::vector<E2> vals(total);
  for (size_t i = 0; i < total; i++) {
    if (opened[2 * i] >= GL_P || opened[2 * i + 1] >= GL_P) return MS_OK;
    vals[i] = E2{opened[2 * i], opened[2 * i + 1]};
  }
  *accepted = pcs_verify_standalone(params_from7(params7), commits, ln, ws, pts, vals, fri, fri_len, ch->ch) ? 1 : 0;
  return MS_OK;
  MS_CATCH
}

int32_t ms_mmcs_open(ms_mmcs* m, size_t index, uint64_t* vals_out, uint8_t* proof_out, size_t* n_siblings) {
  MS_TRY Ctx& ctx = *m->ctx;
  const DTree& t = m->pd().tree;
  unsigned lmh = log2_strict(t.max_height());
  if (index >= t.max_height()) throw std::runtime_error("index out of range");
  std::vector<GatherReq> reqs;
  size_t off = 0;
  for (auto& dm : m->pd().ldes) {
    GatherReq q{dm.d(), dm.h, index >> (lmh - log2_strict(dm.h)), (uint32_t)dm.w, 0, off};
    reqs.push_back(q);
    off += dm.w * 8;
  }
  size_t vals_bytes = off;
  size_t ns = t.cap_layer();
  for (size_t i = 0; i < ns; i++) {
    GatherReq q{t.base() + t.layer_off[i], 0, (index >> i) ^ 1, 1, 1, off};
    reqs.push_back(q);
    off += 32;
  }
  std::vector<uint8_t> g(off);
  gather_rows(ctx, reqs, g.data(), off);
  memcpy(vals_out, g.data(), vals_bytes);
  memcpy(proof_out, g.data() + vals_bytes, ns * 32);
  *n_siblings = ns;
  return MS_OK;
  MS_CATCH
}
void ms_mmcs_destroy(ms_mmcs* m) {
  if (!m) return;
  ms_ctx* c = m->owner;
  ms_system* lender = m->lender;
  m->data = PcsData();
  delete m;
  if (lender) system_unref(lender);
  ctx_unref(c);
}

// ---------------------------------------------------------------- Level 2: the prover's steps on device handles
// For a host that keeps the reference's own prover loop (src/prover.rs:290-603) and calls the device per step: nothing
// but commitments, accumulators, challenges and the final opening crosses the boundary between the calls.
static ms_trace* new_trace(ms_ctx* c, DMat&& m, unsigned log_n, int kind) {
  ms_trace* t = new ms_trace();
  t->owner = c;
  t->ctx = &c->ctx;
  t->m = std::move(m);
  t->log_n = log_n;
  t->kind = kind;
  c->refs++;
  return t;
}
void ms_trace_destroy(ms_trace* t) {
  if (!t) return;
  ms_ctx* c = t->owner;
  t->m = DMat();
  delete t;
  ctx_unref(c);
}
int32_t ms_trace_info(const ms_trace* t, uint64_t out3[3]) {
  MS_TRY if (!t) throw std::runtime_error("null trace handle");
  out3[0] = t->m.h;
  out3[1] = t->m.w;
  out3[2] = (uint64_t)t->kind;
  return MS_OK;
  MS_CATCH
}

static void need_device_witness(const HWitness& w) {
  if (w.host_resident) throw std::runtime_error("the Level-2 entry points take a device-resident witness (ms_witness_create)");
  if (w.has_remote) throw std::runtime_error("this witness lacks traces that another rank computes");
}

int32_t ms_system_preprocessed_mmcs(ms_system* sys, ms_mmcs** out) {
  *out = nullptr;
  MS_TRY HSystem& s = *sys->sys;
  if (!s.has_pre) return MS_OK;  // no preprocessed traces: *out stays null
  std::unique_ptr<ms_mmcs> m(new ms_mmcs());
  m->ctx = s.ctx;
  m->borrowed = &s.pre_data;
  m->lender = sys;
  sys->refs++;
  m->owner = sys->owner;
  sys->owner->refs++;
  *out = m.release();
  return MS_OK;
  MS_CATCH
}

int32_t ms_witness_commit_stage1(ms_witness* w, uint8_t* cap_out, ms_mmcs** out) {
  *out = nullptr;
  MS_TRY HWitness& wit = *w->w;
  HSystem& sys = *wit.sys;
  Ctx& ctx = *sys.ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  need_device_witness(wit);
  std::unique_ptr<ms_mmcs> m(new ms_mmcs());
  m->ctx = &ctx;
  std::vector<DMat> ldes;
  for (size_t ci = 0; ci < sys.circuits.size(); ci++) {
    const size_t h = wit.heights[ci];
    if (!h) continue;
    const size_t wd = sys.circuits[ci].main_width;
    const unsigned logn = log2_strict(h), lb = (unsigned)sys.params.log_blowup;
    DBuf<u64> ev(ctx, h * wd);
    transpose_in(ctx, wit.traces[ci].p, ev.p, h, wd, true);
    DMat lde;
    lde.h = h << lb;
    lde.w = wd;
    lde.buf = DBuf<u64>(ctx, lde.h * wd);
    coset_lde(ctx, ev.p, lde.d(), logn, lb, wd);
    ldes.push_back(std::move(lde));
  }
  if (ldes.empty()) throw std::runtime_error("cannot prove with every circuit deactivated (all traces empty)");
  commit_matrices(ctx, std::move(ldes), (unsigned)sys.params.cap_height, m->data);
  std::vector<Digest> cap = merkle_cap(ctx, m->data.tree);
  for (size_t i = 0; i < cap.size(); i++) memcpy(cap_out + 32 * i, cap[i].b, 32);
  m->owner = w->owner->owner;
  m->owner->refs++;
  *out = m.release();
  return MS_OK;
  MS_CATCH
}

int32_t ms_challenger_observe_claims(ms_challenger* ch, ms_witness* w) {
  MS_TRY HWitness& wit = *w->w;
  Ctx& ctx = *wit.sys->ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  need_device_witness(wit);
  observe_claims(ctx, ch->ch, wit);
  return MS_OK;
  MS_CATCH
}

int32_t ms_witness_claims_accumulator(ms_witness* w, const uint64_t beta[2], const uint64_t gamma[2], uint64_t acc_out[2]) {
  MS_TRY HWitness& wit = *w->w;
  Ctx& ctx = *wit.sys->ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  need_device_witness(wit);
  const size_t n_claims = wit.claim_offsets.size() - 1;
  const E2 b = canonical_e2(beta, "beta"), g = canonical_e2(gamma, "gamma");
  E2 a = n_claims ? claims_accumulator(ctx, wit.d_claim_data.p, wit.d_claim_offsets.p, n_claims, b, g) : e2(0);
  acc_out[0] = a.c0;
  acc_out[1] = a.c1;
  return MS_OK;
  MS_CATCH
}

int32_t ms_stage2_build(ms_witness* w, const uint64_t beta[2], const uint64_t gamma[2], const uint64_t acc_in[2], uint64_t* accs_out,
                        ms_trace** traces_out) {
  MS_TRY HWitness& wit = *w->w;
  HSystem& sys = *wit.sys;
  Ctx& ctx = *sys.ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  need_device_witness(wit);
  const E2 beta_e = canonical_e2(beta, "beta"), gamma_e = canonical_e2(gamma, "gamma");
  E2 acc = canonical_e2(acc_in, "accumulator");
  size_t pos = 0;
  std::vector<std::unique_ptr<ms_trace, void (*)(ms_trace*)>> made;
  for (size_t ci = 0; ci < sys.circuits.size(); ci++) {
    const size_t n = wit.heights[ci];
    if (!n) continue;
    const HCircuit& c = sys.circuits[ci];
    DMat ev;
    ev.h = n;
    ev.w = c.stage2_width;
    ev.buf = DBuf<u64>(ctx, n * c.stage2_width);
    E2 tot;
    {
      DBuf<E2> d_tot(ctx, 1);
      stage2_circuit_async(ctx, sys, wit, ci, beta_e, gamma_e, ev.d(), d_tot.p);
      ctx.d2h(&tot, d_tot.p, sizeof(E2));
    }
    acc = e2_add(acc, tot);
    accs_out[2 * pos] = acc.c0;
    accs_out[2 * pos + 1] = acc.c1;
    made.emplace_back(new_trace(w->owner->owner, std::move(ev), log2_strict(n), 0), ms_trace_destroy);
    pos++;
  }
  for (size_t i = 0; i < made.size(); i++) traces_out[i] = made[i].release();
  return MS_OK;
  MS_CATCH
}

int32_t ms_pcs_commit_traces(ms_ctx* c, uint32_t log_blowup, uint32_t cap_height, size_t n, ms_trace* const* evals, uint8_t* cap_out,
                             ms_mmcs** out) {
  *out = nullptr;
  MS_TRY Ctx& ctx = c->ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  if (log_blowup < 1 || log_blowup > 8) throw std::runtime_error("log_blowup out of range");
  std::unique_ptr<ms_mmcs> m(new ms_mmcs());
  m->ctx = &ctx;
  std::vector<DMat> ldes;
  // every handle is checked before the first one is consumed: a bad handle at position i leaves the others intact
  if (n == 0 || !evals) throw std::runtime_error("ms_pcs_commit_traces: no matrices");
  for (size_t i = 0; i < n; i++) {
    ms_trace* t = evals[i];
    if (!t || t->ctx != &ctx || t->kind != 0 || !t->m.d()) throw std::runtime_error("ms_pcs_commit_traces: not an evaluation handle of this context");
    if (t->log_n > NTT_MAX_LOG || t->log_n + log_blowup > TW_LOG) throw std::runtime_error("matrix too tall");
    for (size_t j = 0; j < i; j++)
      if (evals[j] == t) throw std::runtime_error("ms_pcs_commit_traces: the same handle twice");
  }
  for (size_t i = 0; i < n; i++) {
    ms_trace* t = evals[i];
    DMat lde;
    lde.h = t->m.h << log_blowup;
    lde.w = t->m.w;
    lde.buf = DBuf<u64>(ctx, lde.h * lde.w);
    coset_lde(ctx, t->m.d(), lde.d(), t->log_n, log_blowup, lde.w);  // consumes the evaluations
    t->m = DMat();
    ldes.push_back(std::move(lde));
  }
  commit_matrices(ctx, std::move(ldes), cap_height, m->data);
  std::vector<Digest> cap = merkle_cap(ctx, m->data.tree);
  for (size_t i = 0; i < cap.size(); i++) memcpy(cap_out + 32 * i, cap[i].b, 32);
  m->owner = c;
  c->refs++;
  *out = m.release();
  return MS_OK;
  MS_CATCH
}

int32_t ms_quotient(ms_system* sys, size_t ci, uint32_t log_n, ms_mmcs* s1, size_t s1_idx, ms_mmcs* s2, size_t s2_idx,
                    const uint64_t publics8[8], const uint64_t alpha[2], ms_trace** q_lde_out) {
  *q_lde_out = nullptr;
  MS_TRY HSystem& s = *sys->sys;
  Ctx& ctx = *s.ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  if (ci >= s.circuits.size()) throw std::runtime_error("circuit index out of range");
  const HCircuit& c = s.circuits[ci];
  const unsigned lb = (unsigned)s.params.log_blowup, log_q = log2_strict(c.quotient_degree());
  if (log_n > NTT_MAX_LOG || log_n + lb > TW_LOG) throw std::runtime_error("ms_quotient: log_n out of range");  // (before any 1 << log_n)
  if (!s1 || !s2 || !publics8 || !alpha) throw std::runtime_error("ms_quotient: null argument");
  if (s1->ctx != &ctx || s2->ctx != &ctx) throw std::runtime_error("commitment belongs to another context");
  if (s1_idx >= s1->pd().ldes.size() || s2_idx >= s2->pd().ldes.size()) throw std::runtime_error("matrix index out of range");
  const DMat& m1 = s1->pd().ldes[s1_idx];
  const DMat& m2 = s2->pd().ldes[s2_idx];
  const size_t n = size_t(1) << log_n, nq = n << log_q;
  if (m1.h != (n << lb) || m2.h != (n << lb) || m1.w != c.main_width || m2.w != c.stage2_width)
    throw std::runtime_error("ms_quotient: the committed matrices do not have this circuit's shape");
  QuotientArgs qa;
  if (s.has_pre && s.pre_indices[ci] >= 0) {
    const DMat& pm = s.pre_data.ldes[s.pre_indices[ci]];
    if (pm.h != (n << lb)) throw std::runtime_error("main trace height must equal preprocessed trace height");
    qa.pre = pm.d();
    qa.pre_h = pm.h;
  }
  qa.s1 = m1.d();
  qa.s1_h = m1.h;
  qa.s2 = m2.d();
  qa.s2_h = m2.h;
  qa.log_n = log_n;
  qa.log_q = log_q;
  for (int k = 0; k < 8; k++) {
    if (publics8[k] >= GL_P) throw std::runtime_error("non-canonical public value");
    qa.publics[k] = publics8[k];
  }
  qa.alpha = canonical_e2(alpha, "alpha");
  DBuf<u64> qv(ctx, nq * 2);
  quotient_eval(ctx, c.prog, qa, qv.p);
  DMat lde;
  lde.h = n << lb;
  lde.w = 2 << log_q;
  lde.buf = DBuf<u64>(ctx, lde.h * lde.w);
  quotient_lde(ctx, qv.p, lde.d(), log_n, log_q, lb, 2);
  *q_lde_out = new_trace(sys->owner, std::move(lde), log_n, 1);
  return MS_OK;
  MS_CATCH
}

int32_t ms_pcs_commit_ldes(ms_ctx* c, uint32_t cap_height, size_t n, ms_trace* const* ldes, uint8_t* cap_out, ms_mmcs** out) {
  *out = nullptr;
  MS_TRY Ctx& ctx = c->ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  std::unique_ptr<ms_mmcs> m(new ms_mmcs());
  m->ctx = &ctx;
  std::vector<DMat> ms;
  for (size_t i = 0; i < n; i++) {
    ms_trace* t = ldes[i];
    if (!t || t->ctx != &ctx || t->kind != 1 || !t->m.d()) throw std::runtime_error("ms_pcs_commit_ldes: not an LDE handle of this context");
    ms.push_back(std::move(t->m));  // the commitment takes the matrix over
    t->m = DMat();
  }
  commit_matrices(ctx, std::move(ms), cap_height, m->data);
  std::vector<Digest> cap = merkle_cap(ctx, m->data.tree);
  for (size_t i = 0; i < cap.size(); i++) memcpy(cap_out + 32 * i, cap[i].b, 32);
  m->owner = c;
  c->refs++;
  *out = m.release();
  return MS_OK;
  MS_CATCH
}

int32_t ms_blake3(ms_ctx* c, const uint8_t* bytes, size_t len, uint8_t out32[32]) {
  MS_TRY Ctx& ctx = c->ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  size_t pl = len & 7, nw = len >> 3;
  DBuf<uint8_t> pre(ctx, 8);
  DBuf<u64> words(ctx, std::max<size_t>(nw, 1));
  if (pl) ctx.h2d(pre.p, bytes, pl);
  if (nw) ctx.h2d(words.p, bytes + pl, nw * 8);
  Digest d = blake3_device(ctx, pre.p, pl, words.p, nw);
  memcpy(out32, d.b, 32);
  return MS_OK;
  MS_CATCH
}

int32_t ms_stage2_trace(ms_ctx* c, size_t height, size_t L, const uint64_t* mult, const uint64_t* arg_offsets,
                        const uint64_t* args, const uint64_t beta[2], const uint64_t gamma[2], const uint64_t acc_in[2],
                        uint64_t* trace_out, uint64_t acc_out[2]) {
  MS_TRY Ctx& ctx = c->ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  check_pow2(height);
  DLookups lk;
  lk.height = height;
  lk.num_lookups = L;
  std::vector<uint32_t> offs(L + 1);
  for (size_t i = 0; i <= L; i++) offs[i] = (uint32_t)arg_offsets[i];
  lk.args_width = offs[L];
  lk.arg_offsets = DBuf<uint32_t>(ctx, L + 1);
  ctx.h2d(lk.arg_offsets.p, offs.data(), (L + 1) * 4);
  lk.mult = DBuf<u64>(ctx, std::max<size_t>(height * L, 1));
  lk.args = DBuf<u64>(ctx, std::max<size_t>(height * lk.args_width, 1));
  if (L) ctx.h2d(lk.mult.p, mult, height * L * 8);
  if (lk.args_width) ctx.h2d(lk.args.p, args, height * lk.args_width * 8);
  size_t w2 = std::max<size_t>(L, 1) * 2;
  DBuf<u64> out(ctx, height * w2);
  E2 total = stage2_build(ctx, lk, e2(beta[0], beta[1]), e2(gamma[0], gamma[1]), out.p);
  E2 acc = e2_add(e2(acc_in[0], acc_in[1]), total);
  acc_out[0] = acc.c0;
  acc_out[1] = acc.c1;
  download_rowmajor(ctx, out.p, height, w2, true, trace_out);  // undo the bit-reversed row order
  return MS_OK;
  MS_CATCH
}

int32_t ms_claims_accumulator(ms_ctx* c, size_t n_claims, const uint64_t* offs, const uint64_t* data, const uint64_t beta[2],
                              const uint64_t gamma[2], uint64_t acc_out[2]) {
  MS_TRY Ctx& ctx = c->ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  size_t tot = n_claims ? offs[n_claims] : 0;
  DBuf<u64> d_offs(ctx, n_claims + 1), d_data(ctx, std::max<size_t>(tot, 1));
  ctx.h2d(d_offs.p, offs, (n_claims + 1) * 8);
  if (tot) ctx.h2d(d_data.p, data, tot * 8);
  E2 a = claims_accumulator(ctx, d_data.p, d_offs.p, n_claims, e2(beta[0], beta[1]), e2(gamma[0], gamma[1]));
  acc_out[0] = a.c0;
  acc_out[1] = a.c1;
  return MS_OK;
  MS_CATCH
}

int32_t ms_quotient_values(ms_system* sys, size_t ci, const uint64_t publics8[8], uint32_t log_n, uint32_t log_q,
                           const uint64_t* pre_q, const uint64_t* s1_q, const uint64_t* s2_q, const uint64_t alpha[2],
                           uint64_t* out) {
  MS_TRY HSystem& s = *sys->sys;
  Ctx& ctx = *s.ctx;
  if (ci >= s.circuits.size()) throw std::runtime_error("circuit index out of range");
  const HCircuit& c = s.circuits[ci];
  size_t nq = size_t(1) << (log_n + log_q);
  DBuf<u64> pre, s1, s2;
  if (c.pre_width) pre = upload_colmajor(ctx, pre_q, nq, c.pre_width, true);
  s1 = upload_colmajor(ctx, s1_q, nq, c.main_width, true);
  s2 = upload_colmajor(ctx, s2_q, nq, c.stage2_width, true);
  QuotientArgs qa;
  qa.pre = pre.p;
  qa.s1 = s1.p;
  qa.s2 = s2.p;
  qa.pre_h = qa.s1_h = qa.s2_h = nq;
  qa.log_n = log_n;
  qa.log_q = log_q;
  memcpy(qa.publics, publics8, 64);
  qa.alpha = e2(alpha[0], alpha[1]);
  DBuf<u64> q(ctx, nq * 2);
  quotient_eval(ctx, c.prog, qa, q.p);
  download_rowmajor(ctx, q.p, nq, 2, true, out);
  return MS_OK;
  MS_CATCH
}

int32_t ms_field_op(ms_ctx* c, int32_t op, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {
  MS_TRY Ctx& ctx = c->ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  size_t words = op >= 4 ? 2 * n : n;
  DBuf<u64> da(ctx, words), db(ctx, words), dout(ctx, words);
  ctx.h2d(da.p, a, words * 8);
  if (b) ctx.h2d(db.p, b, words * 8);
  field_op(ctx, op, da.p, b ? db.p : da.p, n, dout.p);
  ctx.d2h(out, dout.p, words * 8);
  return MS_OK;
  MS_CATCH
}

}  // extern "C"
