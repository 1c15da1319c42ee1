// Internal header of the MI355X prover library: device context, pooled device memory, kernel launchers.
// Nothing here crosses the C ABI (include/mstark.h does).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "gl_dev.h"

namespace msamd {

#define HIP_CHECK(expr)                                                                                   \
  do {                                                                                                    \
    hipError_t _e = (expr);                                                                               \
    if (_e != hipSuccess)                                                                                 \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " + __FILE__ + ":" + \
                               std::to_string(__LINE__));                                                 \
  } while (0)

static constexpr unsigned TW_LOG = 28;        // twiddle tables cover exponents of the order-2^28 root (LDE heights up to 2^28)
static constexpr unsigned TW_HALF = 14;       // two-level split: W^e = T1[e >> 14] * T0[e & 16383]
static constexpr unsigned NTT_MAX_LOG = 26;   // largest transform (trace height) supported

struct Digest {
  uint8_t b[32];
};

// kernel classes for which launch durations can be sampled with HIP events (bench.py roofline leg)
enum KernelId : int {
  K_NTT_STRIDED = 0,  // generic LDS radix-2 passes (sizes the register kernels do not cover)
  K_NTT_CONTIG,
  K_NTT12_DIF,        // register radix-16 kernels: 12-bit contiguous pass, 8-bit strided pass
  K_NTT12_DIT,
  K_NTT8S_DIF,
  K_NTT8S_DIT,
  K_LEAF_HASH,
  K_COMPRESS,
  K_STAGE2,
  K_QUOTIENT,
  K_BARY,
  K_DEEP,
  K_FRI_FOLD,
  K_TRANSPOSE,
  K_OTHER,
  K_COUNT
};
const char* kernel_name(int id);

struct KernelStat {
  uint64_t launches = 0;
  double ms = 0;
  double alg_bytes = 0;  // algorithmic bytes moved (each logical input read once, each output written once)
  double units = 0;      // work items other than bytes where a class has them (Poseidon2 permutations of the BabyBear path)
};

struct Ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // bulk uploads of a host-resident witness, running beside the kernels of `stream`
  hipStream_t claims_stream = nullptr;  // the claims of a host-resident witness: DMA copies beside the trace's pulling kernels
  std::vector<hipEvent_t> group_events;  // one per row group of a narrow upload in flight, created on demand (group_event)
  hipEvent_t group_event(size_t i);
  // Side stream for the short circuits of a system (prover.hip): between side_fork() and side_join() the launches queued
  // inside a SideScope go to `side_stream` (the scope swaps `stream`) and run beside the long kernels of the main stream
  // instead of in front of them. Blocks allocated inside a scope come from a pool of their own (`pool_free_side`), and a
  // release of such a block while the streams are forked is deferred to the join, so no block ever changes streams
  // without an event in between.
  hipStream_t main_stream = nullptr, side_stream = nullptr;
  hipEvent_t side_ev[2] = {nullptr, nullptr};
  bool side_forked = false;
  int side_depth = 0;
  bool side_enabled = true;
  unsigned side_max_log = 12;  // circuits of at most 2^side_max_log rows count as short (MSAMD_SIDE_MAX_LOG)
  std::multimap<size_t, void*> pool_free_side;
  std::map<void*, size_t> side_live;                      // live blocks of the side pool
  std::vector<std::pair<size_t, void*>> side_deferred;    // released while forked
  void side_config(); // read MSAMD_NO_SIDE_STREAM / MSAMD_SIDE_MAX_LOG (at the start of every proof: tests flip them)
  void side_fork();   // the side stream waits for everything queued on the main stream so far
  void side_join();   // the main stream waits for everything queued on the side stream so far
  unsigned side_delay_us = 0, main_delay_us = 0;  // MSAMD_SIDE_DELAY_US / MSAMD_MAIN_DELAY_US (diagnostics)
  unsigned copy_delay_us = 0;                     // MSAMD_COPY_DELAY_US (diagnostics)
  void copy_delay();  // delays the copy stream by copy_delay_us (called where a proof's uploads begin)
  void join_side_for_copy();  // in front of a read-back issued at once on the main stream (its source may be the side stream's work)
  hipEvent_t copy_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  u64 *tw0 = nullptr, *tw1 = nullptr, *tw0i = nullptr, *tw1i = nullptr;
  // compact per-order tables: table r (root of order 2^r, r = 1..12) holds w^i for i < 2^(r-1) at offset 2^(r-1) - 1
  u64 *twc = nullptr, *twci = nullptr;
  // cubes for the radix-4 butterflies: table r (r = 2..12) holds w^(3i) for i < 2^(r-2) at offset 2^(r-2) - 1
  u64 *twc3 = nullptr, *twc3i = nullptr;
  // full per-order tables: table r (r = 1..12) holds w^i for ALL i < 2^r at offset 2^r - 1 (the radix-16 rounds in
  // "16-point network, then one twiddle per value" form need exponents up to 15/16 of the order)
  u64 *twf = nullptr, *twfi = nullptr;
  u32* tree_counter = nullptr;  // ticket counter of the in-launch sub-tree hand-over (hash.hip): zero between launches
  // pooled device memory: exact-size buckets, reused across proofs
  std::multimap<size_t, void*> pool_free;
  std::map<void*, size_t> pool_live;
  size_t pool_bytes = 0;
  int fail_alloc_countdown = 0;  // diagnostics (ms_ctx_debug_fail_alloc): the n-th alloc from now throws
  // pinned staging for small transfers: the first half takes uploads (bump-allocated; every block stays untouched
  // until the stream has been synchronised, then the half is reused), the second half receives read-backs.
  // Pageable buffers would make hipMemcpyAsync stage and block on the host for every call.
  uint8_t* pinned = nullptr;
  size_t pinned_half = 0, up_used = 0, down_used = 0;
  // Transfers that do not fit the staging halves (a whole witness at creation, a matrix read back by a PCS-level call) go through
  // this page-locked bounce buffer chunk by chunk, each chunk waited for: the caller's pageable memory is never handed to an
  // asynchronous copy (the runtime would page-lock it on the fly and let go of that lock at a time of its own choosing - next
  // to hipHostRegister / hipHostUnregister of recycled heap addresses that ended in a GPU memory access fault on a host
  // address in round 4's fuzzing)
  uint8_t* bounce = nullptr;
  static constexpr size_t BOUNCE_BYTES = size_t(4) << 20;
  void bounce_h2d(void* dst, const void* src, size_t n, hipStream_t on = nullptr);  // on: another stream than `stream`
  void bounce_d2h(void* dst, const void* src, size_t n);
  struct PendingD2H {
    void* dst;
    const void* src;
    size_t off, n;
    bool copied;  // a hipMemcpyAsync into the staging buffer has already been issued
  };
  std::vector<PendingD2H> down_pending;
  bool down_direct = false;        // some queued read-back bypasses the flag-copy kernel
  uint8_t* pinned_dev = nullptr;   // device-visible address of `pinned`
  uint32_t *flag_host = nullptr, *flag_dev = nullptr;  // completion flag of the flag-copy kernel (in the pinned block)
  uint32_t flag_seq = 0;
  // LDE scale vectors (g w^k0)^j / n, per log_n and log_blowup
  std::map<std::pair<unsigned, unsigned>, u64*> lde_scales;
  // profiling
  uint32_t prof_mask = 0;
  struct Pending {
    int id;
    hipEvent_t a, b;
    double bytes;
  };
  std::vector<Pending> prof_pending;
  std::vector<hipEvent_t> event_pool;
  KernelStat stats[K_COUNT];

  // Progress of the multi-rank prover's exchanges (prover_sharded.inc): which collective this rank entered last, how many
  // it has entered, and whether it has returned from it. Read by another thread (ms_ctx_comm_progress: a watchdog that must
  // say WHERE a joint proof hangs when a peer never arrives).
  std::mutex comm_mu;
  std::string comm_what;
  uint64_t comm_seq = 0;
  uint64_t host_syncs = 0;  // how often the host has waited for this context's stream (ms_ctx_sync_count)
  bool comm_in_flight = false;

  explicit Ctx(int dev);
  ~Ctx();
  void* alloc(size_t bytes);
  void release(void* p);
  void trim();  // return pooled blocks to the driver
  void sync() { sync_and_deliver(); }
  void sync_and_deliver();  // stream synchronisation + hand-over of queued read-backs
  void h2d(void* dst, const void* src, size_t n);
  void d2h(void* dst, const void* src, size_t n);        // synchronous (waits for the stream)
  void d2h_queue(void* dst, const void* src, size_t n);  // delivered to dst by the next d2h / sync_and_deliver
  // the same without the final host copy: the bytes can be read at the returned address of the pinned staging buffer after
  // the next synchronisation and until the next read-back is queued (nullptr: does not fit, use d2h_queue)
  const uint8_t* d2h_queue_staged(const void* src, size_t n);
  std::vector<uint8_t> host_scratch2;  // the joint prover's query openings (prover_sharded.inc)
  std::vector<uint8_t> host_scratch;  // grows once; large per-proof host buffers that would otherwise be page-faulted in anew
  const u64* lde_scale(unsigned log_n, unsigned log_blowup);
  // profiling hooks around one launch
  bool prof_on(int id) const { return (prof_mask >> id) & 1u; }
  hipEvent_t prof_begin(int id);
  void prof_end(int id, hipEvent_t a, double bytes);
  void prof_collect();
};

// Ticket counter of the in-launch sub-tree hand-over for the stream the next launch goes to: the main and the side stream run
// concurrently, and two trees sharing one counter would mix their tickets (a wrong "last arriver"). One slot per stream, a
// cache line apart. (The hand-over itself keeps the protocol of MI355X_MICROARCH.md, "inter-workgroup visibility": sc1
// write-through payload, drained, relaxed agent-scope ticket, sc1 loads by the last arriver - a release / acquire pair at
// agent scope would write back and invalidate the whole L2 of the XCD for 32 bytes of payload.)
inline u32* tree_counter_slot(Ctx& ctx) { return ctx.tree_counter + (ctx.side_depth > 0 ? 32 : 0); }

// roctx range with one of the reference's tracing span names (src/prover.rs:289,336,391,413,437,538: "stark/prove",
// "stark/stage1_commit", ...), so that rocprofv3 --marker-trace shows the same phases the reference's tracing subscriber
// does. The roctx library is looked up at run time; without it the ranges cost nothing.
struct RoctxRange {
  bool on = false;
  explicit RoctxRange(const char* name);
  ~RoctxRange();
  void next(const char* name);  // close the current range and open another one
};

// launches, copies and allocations inside the scope go to the side stream (no-op unless the streams are forked)
struct SideScope {
  Ctx& ctx;
  bool on;
  SideScope(Ctx& c, bool use) : ctx(c), on(use && c.side_forked) {
    if (on) {
      ctx.side_depth++;
      ctx.stream = ctx.side_stream;
    }
  }
  ~SideScope() {
    if (on && --ctx.side_depth == 0) ctx.stream = ctx.main_stream;
  }
};

// drop read-backs queued by the calling thread whose destinations an error has unwound (called by the C-ABI catch blocks)
// Page-locking of caller memory for host-resident witnesses (both configurations), counted per range process-wide: two witnesses
// made from the same buffers share one lock, and the range keeps it until the LAST of them is gone - the second
// hipHostRegister of a range only reports "already registered", and a plain unregister by the first owner used to leave the
// second one's uploads reading memory the GPU could no longer see. Returns 1 = locked (by this call or an earlier one of the
// library), 2 = the application had registered the range itself (left alone), 0 = could not be locked.
int host_range_pin(const void* p, size_t bytes);
void host_range_unpin(const void* p);  // for a range host_range_pin returned 1 for
void abandon_pending();

// RAII device buffer from the pool
template <class T>
struct DBuf {
  Ctx* ctx = nullptr;
  T* p = nullptr;
  size_t n = 0;
  DBuf() {}
  DBuf(Ctx& c, size_t count) : ctx(&c), n(count) { p = count ? (T*)c.alloc(count * sizeof(T)) : nullptr; }
  DBuf(const DBuf&) = delete;
  DBuf& operator=(const DBuf&) = delete;
  DBuf(DBuf&& o) noexcept : ctx(o.ctx), p(o.p), n(o.n) { o.p = nullptr; }
  DBuf& operator=(DBuf&& o) noexcept {
    if (this != &o) {
      reset();
      ctx = o.ctx;
      p = o.p;
      n = o.n;
      o.p = nullptr;
    }
    return *this;
  }
  void reset() {
    if (p && ctx) ctx->release(p);
    p = nullptr;
  }
  ~DBuf() { reset(); }
};

// column-major device matrix: element (r, c) at d[c * h + r]
struct DMat {
  DBuf<u64> buf;
  size_t h = 0, w = 0;
  u64* d() const { return buf.p; }
};

// ---------------------------------------------------------------- ntt.hip
// In-place batched transforms over `ncols` contiguous columns of length 2^logn (column c at data + c * 2^logn).
//   ntt_dif: natural-order input  -> bit-reversed-order output (storage row r holds frequency bitrev(r))
//   ntt_dit: bit-reversed input   -> natural-order output
// inverse = use w^-1 twiddles (no 1/n scaling unless out_mul is given). `src`/`scale` (optional) let the first
// pass read from another buffer with a per-row factor: dst[c][r] <- transform(src[c_src][r] * scale[r]).
struct NttSrc {
  const u64* src = nullptr;      // column c read at src + (c / src_div) * 2^logn  (src_div copies per source column)
  unsigned src_div = 1;
  const u64* scale = nullptr;    // per destination column group: scale + (c % src_div) * 2^logn
};
void ntt_dif(Ctx& ctx, u64* data, unsigned logn, size_t ncols, bool inverse, const NttSrc* from = nullptr, u64 out_mul = 1);
void ntt_dit(Ctx& ctx, u64* data, unsigned logn, size_t ncols, bool inverse, u64 out_mul = 1, bool first_pass_done = false);
// the first (4096-row tile) pass of ntt_dit on one of eight row groups; ntt_dit(.., first_pass_done = true) finishes (ntt.hip)
void ntt_dit_first_pass_part(Ctx& ctx, u64* data, unsigned logn, size_t ncols, bool inverse, unsigned part_rev);
void transpose_in_rows_part(Ctx& ctx, const u64* rowmajor, u64* colmajor, size_t h, size_t w, unsigned part);
void pull_widen_runs(const uint8_t* host_packed, unsigned bytes, size_t count, u64* out, size_t run_words, size_t run_stride, hipStream_t stream);
// row-major host layout (h x w) on device -> column-major, optionally with rows bit-reversed
void transpose_in(Ctx& ctx, const u64* rowmajor, u64* colmajor, size_t h, size_t w, bool bitrev_rows);
void transpose_out(Ctx& ctx, const u64* colmajor, u64* rowmajor, size_t h, size_t w, bool bitrev_rows);
// out[i] = the i-th little-endian `bytes`-byte value of `packed` (bytes = 1, 2, 4), on `stream`
void widen_words(const uint8_t* packed, unsigned bytes, size_t count, u64* out, hipStream_t stream);
// the same from PINNED HOST memory, read by the kernel itself (no staging copy): host_packed 16-byte aligned
void pull_widen_words(const uint8_t* host_packed, unsigned bytes, size_t count, u64* out, hipStream_t stream);
// coefficients (unscaled inverse DFT output, natural order, column-major n x w) -> bit-reversed coset LDE (Bn x w)
void lde_from_coeffs(Ctx& ctx, const u64* coef, u64* lde, unsigned logn, unsigned log_blowup, size_t w);
// evaluations in bit-reversed row order (column-major n x w, destroyed) -> bit-reversed coset LDE (Bn x w)
void coset_lde(Ctx& ctx, u64* evals_bitrev, u64* lde, unsigned logn, unsigned log_blowup, size_t w, bool first_pass_done = false);
// src/prover.rs:631-717 fused: quotient values in storage (bit-reversed) order, nq x D column-major (destroyed)
// -> committed quotient LDE (B n x qD)
void quotient_lde(Ctx& ctx, u64* qvals_bitrev, u64* lde, unsigned logn, unsigned logq, unsigned log_blowup, size_t D);

// ---------------------------------------------------------------- hash.hip
struct MatRef {
  const u64* d;
  uint32_t w;
  uint32_t pad;
  uint64_t stride;  // elements between consecutive columns (= the matrix height unless the rows are a window of a taller matrix)
};
// Merkle tree over column-major matrices (heights powers of two), p3 MerkleTreeMmcs semantics
struct DTree {
  std::vector<const u64*> mat_d;   // matrices in input order (not owned)
  std::vector<size_t> mat_h, mat_w;
  std::vector<size_t> mat_stride;  // optional: column stride of each matrix when it is not its height (a row range read in place)
  DBuf<Digest> digests;            // all layers back to back, leaf layer first
  Digest* ext = nullptr;           // layers living in someone else's buffer (FRI tail rounds); overrides `digests`
  Digest* base() const { return ext ? ext : digests.p; }
  std::vector<size_t> layer_off, layer_len;
  unsigned cap_height = 0;
  size_t max_height() const { return layer_len.empty() ? 0 : layer_len[0]; }
  size_t cap_layer() const {
    size_t L = layer_len.size();
    size_t ch = cap_height < L - 1 ? cap_height : L - 1;
    return L - 1 - ch;
  }
};
void merkle_build(Ctx& ctx, DTree& t);                     // fills digests for t.mat_* (already set)
void merkle_alloc(Ctx& ctx, DTree& t, size_t max_height);  // layer table + digest storage only
struct FriChallenge;
void merkle_compress_plain(Ctx& ctx, DTree& t, const FriChallenge* fc = nullptr);            // layers 1.. from a filled leaf layer, no injection
std::vector<Digest> merkle_cap(Ctx& ctx, const DTree& t);  // D2H of the cap layer (synchronises)
// levels above n sub-tree roots, on the device: d_out[0 .. n) = roots, then n / 2 parents, ..., the root (2 n - 1 digests)
void merkle_tree_top(Ctx& ctx, const Digest* d_roots, size_t n, Digest* d_out);
// BLAKE3 of the byte stream prefix (prefix_len bytes) || nwords little-endian u64 words; result to host
Digest blake3_device(Ctx& ctx, const uint8_t* d_prefix, size_t prefix_len, const u64* d_words, size_t nwords);
// the same in two steps, so that ranks can split the chunk range: chaining values of chunks [c0, c1) into cvs[c0..c1),
// then the tree over all nchunks values (overwrites cvs)
size_t blake3_num_chunks(size_t prefix_len, size_t nwords);
void blake3_chunk_cvs(Ctx& ctx, const uint8_t* d_prefix, size_t prefix_len, const u64* d_words, size_t nwords, size_t c0, size_t c1,
                      Digest* cvs);
Digest blake3_from_cvs(Ctx& ctx, Digest* cvs, size_t nchunks);
void blake3_from_cvs_async(Ctx& ctx, Digest* cvs, size_t nchunks, Digest* out_dev);  // launches only; digest left on the device
// a stream whose chunk 0 is known last (it holds a commitment): everything that does not depend on it early, one wave late
struct ChallengeBG;  // lookup_params.h
struct LateChunk0 {
  DBuf<Digest> levels;             // every tree level above the chunks, one after the other
  std::vector<const Digest*> sib;  // per level: the sibling of the leftmost path
  void reset() {
    levels.reset();
    sib.clear();
  }
};
void blake3_late_chunk0_prepare(Ctx& ctx, const Digest* cvs, size_t nchunks, LateChunk0& lc);
void blake3_late_chunk0_finish(Ctx& ctx, const uint8_t* d_prefix, size_t prefix_len, const u64* d_words, size_t nwords, const Digest* d_cap,
                               size_t cap_off, size_t ncap, const LateChunk0& lc, size_t nchunks, Digest* out_dev,
                               ChallengeBG* d_bg = nullptr /* when set: beta and gamma are sampled in the same launch (outer.hip) */,
                               u32* d_state12 = nullptr);

// ---------------------------------------------------------------- lookup.hip
struct JitKernel;
// device-resident LookupValues of one circuit (row-major as the reference stores them)
struct DLookups {
  size_t height = 0, num_lookups = 0, args_width = 0;
  DBuf<u64> mult, args;
  DBuf<uint32_t> arg_offsets;  // num_lookups + 1
};
// stage-2 trace of one circuit: writes column-major (n x max(L,1)*2) with rows bit-reversed (ready for the
// inverse DIT); returns the circuit's total contribution sum_{r,j} mult/msg
E2 stage2_build(Ctx& ctx, const DLookups& lk, E2 beta, E2 gamma, u64* out_colmajor_bitrev, const JitKernel* jit = nullptr);
// launches only: the contribution is left in *total_dev (device memory)
void stage2_build_async(Ctx& ctx, const DLookups& lk, E2 beta, E2 gamma, u64* out_colmajor_bitrev, E2* total_dev,
                        const JitKernel* jit = nullptr);
// the same three with beta / gamma already in device memory (lookup_params.h::ChallengeBG; written by h2d or by the device
// transcript): the forms the by-value ones wrap
struct ChallengeBG;
DBuf<ChallengeBG> challenge_bg_upload(Ctx& ctx, E2 beta, E2 gamma);
void stage2_build_dyn(Ctx& ctx, const DLookups& lk, const ChallengeBG* ch, u64* out_colmajor_bitrev, E2* total_dev, const JitKernel* jit = nullptr);
void stage2_from_trace_dyn(Ctx& ctx, const JitKernel& trace_jit, const u64* d_trace, const u64* d_pre, size_t n, size_t num_lookups,
                           size_t args_width, const ChallengeBG* ch, u64* out_colmajor_bitrev, E2* total_dev);
void claims_accumulator_dyn(Ctx& ctx, const u64* d_claim_data, const u64* d_claim_offsets, size_t n_claims, const ChallengeBG* ch, E2* out_dev);
// the same from the row-major trace (and preprocessed trace) with the circuit's fused kernel: no LookupValues needed
void stage2_from_trace_async(Ctx& ctx, const JitKernel& trace_jit, const u64* d_trace, const u64* d_pre, size_t n, size_t num_lookups,
                             size_t args_width, E2 beta, E2 gamma, u64* out_colmajor_bitrev, E2* total_dev);
void claims_accumulator_async(Ctx& ctx, const u64* d_claim_data, const u64* d_claim_offsets, size_t n_claims, E2 beta, E2 gamma,
                              E2* out_dev);
E2 claims_accumulator(Ctx& ctx, const u64* d_claim_data, const u64* d_claim_offsets, size_t n_claims, E2 beta, E2 gamma);
// the claims part of the transcript as u64 words: count, then per claim its length and elements
// (src/prover.rs:369-373); d_words must hold 1 + n_claims + total_elems words; returns that count
size_t claims_transcript_words(Ctx& ctx, const u64* d_claim_data, const u64* d_claim_offsets, size_t n_claims,
                               size_t total_elems, u64* d_words);

// the same for claims [first, first + count) only: d_offs_first points at the (absolute) offset of claim `first`; d_data_abs and
// d_words_abs are addressed absolutely, i.e. shifted back by the start of the slice the caller holds
void claims_transcript_words_slice(Ctx& ctx, const u64* d_data_abs, const u64* d_offs_first, size_t first, size_t count, size_t n_total,
                                   u64* d_words_abs);

// ---------------------------------------------------------------- quotient.hip
// a circuit's quotient kernel compiled with hiprtc at System::new (quotient_jit.hip); empty = use the interpreter
struct JitKernel {
  void* module = nullptr;    // hipModule_t
  void* function = nullptr;  // hipFunction_t
  bool inline_tables = false;  // quotient kernel: reads zh / zh_inv / alpha powers from the argument block
  unsigned groups = 0;         // stage-2 terms kernel: waves per workgroup, one per group of 16 lookups (0 = a thread does the whole row)
  JitKernel() {}
  JitKernel(const JitKernel&) = delete;
  JitKernel& operator=(const JitKernel&) = delete;
  JitKernel(JitKernel&& o) noexcept : module(o.module), function(o.function), inline_tables(o.inline_tables), groups(o.groups) {
    o.module = o.function = nullptr;
  }
  JitKernel& operator=(JitKernel&& o) noexcept {
    std::swap(module, o.module);
    std::swap(function, o.function);
    std::swap(inline_tables, o.inline_tables);
    std::swap(groups, o.groups);
    return *this;
  }
  ~JitKernel();
};
struct DProgram {
  // register-allocated straight-line program for one circuit (see quotient.hip)
  JitKernel jit;
  DBuf<uint32_t> code;      // 4 words per instruction
  DBuf<u64> consts;
  size_t n_instr = 0, n_slots = 0;
  DBuf<uint32_t> zero_slots;     // slot of each user constraint root
  DBuf<uint32_t> lookup_slots;   // per lookup: mult slot, nargs, arg slots...
  size_t n_zeros = 0, n_lookups = 0, constraint_count = 0;
  size_t main_w = 0, pre_w = 0, s2_w = 0;
  // the same program scheduled by dependency level for ONE WAVE PER ROW (quotient.hip::quotient_wave_k; large programs
  // only): 64 nodes of a level per step, slot = position, levels padded to whole steps
  DBuf<uint32_t> wave_code;      // 4 words per position: kind, 1 on the last step of a level, operand a, operand b
  DBuf<uint32_t> wave_zero_pos;  // position of each user constraint root
  DBuf<uint32_t> wave_lookups;   // per lookup: mult position, nargs, arg positions...
  DBuf<uint32_t> wave_lookup_off;  // where each lookup starts in wave_lookups
  size_t wave_steps = 0, wave_leaf_steps = 0;
};
struct QDyn;  // quotient_params.h: the challenge-dependent part of a quotient launch, in device memory
struct QuotientArgs {
  const u64 *pre = nullptr, *s1 = nullptr, *s2 = nullptr;  // column-major LDEs (bit-reversed rows)
  size_t pre_h = 0, s1_h = 0, s2_h = 0;                    // LDE heights (column strides)
  unsigned log_n = 0, log_q = 0;
  u64 publics[8];   // [beta, gamma, acc_in, acc_out] and alpha when the host knows them ...
  E2 alpha;
  // ... or the same already in device memory, written by the device transcript (outer.hip): the challenge block and the
  // reversed alpha powers (constraint_count entries); publics / alpha above are then ignored
  const QDyn* dyn = nullptr;
  const E2* alpha_rev = nullptr;
};
u64 quotient_inj_norm(unsigned log_n);  // 1 / (n g): the normalisation of the accumulator's injection (src/prover.rs:782-784)

// ---- the outer transcript on the device (outer.hip): beta/gamma, alpha, zeta sampled in the stream; the host replays and checks
struct OuterTarget {  // where one active circuit's challenge-dependent quotient inputs go
  QDyn* dyn;
  E2* alpha_rev;
  unsigned log_n;
  size_t k;  // constraint count
};
bool outer_fits(size_t ncap, size_t na);  // every transcript piece is a single BLAKE3 chunk
void outer_beta_gamma(Ctx& ctx, const Digest* d_digest, ChallengeBG* d_bg, u32* d_state12);
void outer_alpha(Ctx& ctx, const u32* d_state12, const Digest* d_cap, size_t ncap, const E2* d_tot, size_t na, const ChallengeBG* d_bg,
                 const std::vector<OuterTarget>& targets, E2* d_accs, E2* d_alpha, u32* d_state8, DBuf<uint8_t>& keep);
void outer_zeta(Ctx& ctx, const u32* d_state8, const Digest* d_cap, size_t ncap, const u32* d_lds, size_t n_ld, E2* d_points,
                u32* d_state_out = nullptr /* 8 words: the challenger's input buffer behind zeta (where the opened values are absorbed) */);
// joint proof: d_all = world rows of [circuit totals | claims share] (one all_gather) -> d_tot[0] = claims' sum, d_tot[1 + pos] = circuit totals
void outer_joint_totals(Ctx& ctx, const E2* d_all, size_t world, size_t na, const std::vector<int>& src_rank, E2* d_tot);
// writes quotient values in storage order: out[c * nq + t] for c in {0,1}
void quotient_eval(Ctx& ctx, const DProgram& prog, const QuotientArgs& a, u64* out);

// ---------------------------------------------------------------- open.hip
// 1/(z - x_i) for i < H over the bit-reversed coset x_i = 7 w_H^{bitrev(i)}; out: E2[H] (AoS)
// xout (nullable): x_i / (z - x_i) for i < n_x, the barycentric weights of the trace-domain coset
void inv_denoms(Ctx& ctx, E2 z, unsigned log_h, E2* out, E2* xout = nullptr, size_t n_x = 0);
// storage rows [row0, row0 + rows) only (out / xout still point at row 0 of the full arrays)
void inv_denoms_rows(Ctx& ctx, E2 z, unsigned log_h, E2* out, E2* xout, size_t n_x, size_t row0, size_t rows);
void inv_denoms_rows_dev(Ctx& ctx, const E2* z_dev, unsigned log_h, E2* out, E2* xout, size_t n_x, size_t row0, size_t rows);  // point in device memory
void inv_denoms_dev(Ctx& ctx, const E2* z_dev, unsigned log_h, E2* out, E2* xout = nullptr, size_t n_x = 0);  // the point read from device memory
// opened values of a column-major matrix at up to two points: y_p[c] = scale_p * sum_{i<h} col_c[i] * x_i * invden_p[i].
// bary_sums_async only launches (raw sums to device memory, index c * np + p); bary_finish applies scale_p on the host.
// second_is_next: the second point is the first times the generator of the matrix's trace domain (the usual pair zeta,
// zeta * g): its weights are the first point's read through a permutation (open.hip::rev_dec) and xden1 is ignored
void bary_sums_async(Ctx& ctx, const u64* mat, size_t mat_h, size_t w, unsigned log_h, const E2* xden0, const E2* xden1,
                     int npoints, E2* out_dev, bool second_is_next = false);
struct BarySpec {  // one matrix of a batched launch: the arguments of bary_sums_async
  const u64* mat;
  size_t mat_h, w;
  unsigned log_h;
  const E2 *xden0, *xden1;
  int npoints;
  E2* out_dev;
  bool second_is_next;
};
void bary_sums_batch(Ctx& ctx, const std::vector<BarySpec>& specs, DBuf<E2>& partial_keep);  // all matrices in one pair of launches
void bary_finish(const E2* sums, size_t w, unsigned log_h, const E2* zs, int npoints, E2* out /* p * w + c */);
struct DeepMat {
  const u64* d;         // column-major LDE
  uint32_t w;
  uint32_t npoints;
  uint32_t pt[2];       // which of the launch's points each opening uses
  E2 coeff[2];          // alpha^{offset_p}
  uint64_t coeff7[2];   // 7 * coeff.c1 (X^2 = 7), for the lazy product with the column sums
  uint64_t stride;      // column stride in elements; 0 = the launch's height (the rows are the whole matrix)
};
struct DeepPoints {
  uint32_t n;           // opening points at this height (at most two: zeta and zeta * g)
  uint32_t pad;
  const E2* den[2];     // 1 / (z_q - x_i), device
  E2 K[2];              // sum over matrices opened at z_q of coeff * (sum_c alpha^c y_q[c])
  uint32_t shift[2];    // 0, or: z_q = z' * w^shift for the point z' whose denominators den[q] holds (w = the domain's generator);
                        // the kernel then reads den[q] through open.hip::rev_dec, and K / the coefficients carry the factor w^-shift
};
// ---- the opened values' transcript step on the device (open.hip::open_alpha_k; src/prover.rs:540-580 -> p3 TwoAdicFriPcs::open):
// finish the barycentric sums, absorb every opened value into the transcript (BLAKE3 of state || values), sample the FRI batching
// challenge alpha and fill what the reduced openings need from it - alpha's powers, every matrix's coefficients, every height's
// constants K - so that nothing between the barycentric sums and the end of FRI waits for the host. The host replays the step
// from the raw sums afterwards (it is the authority on the challenge; the values are the kernels' output either way).
struct OpenEntry {      // one (matrix, point) of the opening, in the transcript's observe order (round -> matrix -> point)
  uint32_t sum_off;     // the matrix's raw sums start here in the sums array (value of column c, point p at sum_off + c * np + p)
  uint32_t out_off;     // where this entry's w finished values go in the opened array (= observe order)
  uint32_t w, np, p;    // columns; points of the matrix; which of them this entry is
  uint32_t log_h;       // log2 of the trace height (the barycentric domain: the coset of 2^log_h points)
  uint32_t point_id;    // the point's index in the device point array (zeta, zeta * g ...)
  uint32_t mat;         // index of the matrix's DeepMat in the blob; ~0 = no reduced opening for it
  uint32_t exp;         // its coefficient is alpha^exp ...
  uint32_t slot;        // ... and coeff * sum_c alpha^c y_c is added to K[slot]
  uint64_t s_pow, dinv; // 7^(2^log_h) and 1 / (2^log_h * s_pow): the finishing factor is (z^(2^log_h) - s_pow) * dinv
  uint64_t cmul;        // the coefficient's extra factor (1, or g^-1 for a point read through another one's denominators)
};
struct OpenAlphaArgs {
  const OpenEntry* entries;
  uint32_t n_entries, n_vals, gw, n_slots;
  const E2* sums;       // raw barycentric sums
  const E2* points;     // device opening points
  const uint32_t* state_in;  // 8 words: the challenger's input buffer (the digest its last sample left)
  E2* opened;           // n_vals finished values in observe order
  E2* apow;             // gw + 1 powers of alpha
  struct DeepMat* mats; // the DeepMat blob: coeff / coeff7 are filled here
  E2* K;                // n_slots constants
  uint32_t* state_out;  // 8 words: the input buffer after the sample (where FRI's transcript goes on)
  E2* alpha_out;
  Digest* cv_scratch;   // one chaining value per 1024 bytes of transcript
};
void open_alpha(Ctx& ctx, const OpenAlphaArgs& a);
// ro[i] = sum over matrices/points of coeff * (red_z - sum_c alpha^c m[i][c]) / (z - x_i)
// alpha_pows_host (optional): the same powers on the host; short lists then travel inside the kernel's argument block
void deep_reduce(Ctx& ctx, const std::vector<DeepMat>& mats, const DeepPoints& pts, size_t height, const E2* alpha_pows_dev, E2* ro,
                 const E2* alpha_pows_host = nullptr, Digest* fri_leaves = nullptr /* height / 2 leaf digests of FRI's first round */,
                 const DeepMat* mats_dev = nullptr /* the list already in device memory */,
                 size_t row0 = 0, size_t full_height = 0 /* rows [row0, row0 + height) of a domain of full_height rows (0 = height): pts.den
                                                            are indexed by the FULL domain's row */,
                 const E2* K_dev = nullptr /* the points' constants K in device memory (open_alpha_k) instead of pts.K */);
// FRI: leaves of pairs -> digests handled by merkle_build on a 4-column view; fold:
// row0 / rows_total: `cur`, `roll_in`, `out` are the slice [row0, row0 + rows) of a folded layer of rows_total rows (0 = whole layer)
void fri_fold(Ctx& ctx, const E2* cur, size_t rows, E2 beta, const E2* roll_in /*nullable*/, E2* out, size_t row0 = 0, size_t rows_total = 0);
// Merkle tree of one FRI layer: leaf i = BLAKE3 of the 32-byte row (cur[2i], cur[2i+1]) (ExtensionMmcs flattening)
// With fc, the launch that produces the root also runs the challenger step of that round (see challenge_dev.h);
// cur == nullptr means the leaf layer of t (already allocated) was written by fri_fold_dev.
// log_arity: the round commits rows of 2^log_arity values (FriParameters::max_log_arity, src/types.rs:189-190); a row is hashed
// as one BLAKE3 chunk, which bounds the arity at 64
static const unsigned FRI_MAX_LOG_ARITY = 6;
void fri_tree_build(Ctx& ctx, DTree& t, const E2* cur, size_t rows, const FriChallenge* fc = nullptr, unsigned log_arity = 1);
// gather: rows of column-major matrices and digest siblings for the query phase
struct GatherReq {
  const void* base;   // matrix (u64) or digest layer
  uint64_t stride;    // column stride (elements) for matrices
  uint64_t index;     // row index / digest index
  uint32_t count;     // number of columns (matrices) or 1 (digests)
  uint32_t kind;      // 0 = matrix row (u64 x count), 1 = digest (32 bytes)
  uint64_t out_off;   // byte offset in the output buffer
};
void gather_rows(Ctx& ctx, const std::vector<GatherReq>& reqs, uint8_t* host_out, size_t out_bytes);
// query-phase gather: the same segment list is applied to every query index on the device
struct GatherSeg {
  const void* base;   // matrix (u64, column-major), digest layer, or FRI layer (E2)
  uint64_t stride;    // column stride for matrices
  uint32_t count;     // columns of a matrix row; ignored otherwise
  uint32_t kind;      // 0 = matrix row, 1 = digest, 2 = one E2 value, 3 = `count` consecutive u64 words at element * count
  uint32_t shift;     // element index = (query_index >> shift) ^ flip
  uint32_t flip;
  uint64_t out_off;   // byte offset inside one query's block
};
void gather_queries(Ctx& ctx, const std::vector<GatherSeg>& segs, const std::vector<uint64_t>& indices, size_t bytes_per_query,
                    uint8_t* host_out);
// launches only: segment list uploaded to segs_dev, indices read from device memory, openings left in out_dev
// owner != ~0: only queries with (index >> owner_shift) == owner are served, the other blocks of out_dev stay untouched
void gather_queries_launch(Ctx& ctx, const std::vector<GatherSeg>& segs, GatherSeg* segs_dev, const u64* indices_dev, size_t n_queries,
                           size_t bytes_per_query, uint8_t* out_dev, unsigned owner_shift = 0, uint32_t owner = ~uint32_t(0));
// query-phase challenger step on the device (one-coefficient final polynomial): out_dev[0] = PoW witness,
// out_dev[1 + q] = query index q; state_dev is the challenger state the commit phase left
void fri_query_challenge(Ctx& ctx, const uint32_t* state_dev, const E2* final_dev, unsigned pow_bits, uint32_t n_queries,
                         unsigned log_max_height, u64* out_dev);
// proof-of-work search on the device (single-chunk transcripts); false = not applicable, use the host loop
bool grind_device(Ctx& ctx, const std::vector<uint8_t>& input, unsigned bits, u64* witness_out);
// Last FRI rounds (vectors of <= 2048 elements) in ONE single-workgroup launch: per round leaf hashes, tree, the
// challenger step (observe root, grind, sample beta) and the fold all stay on the device; the host replays the
// transcript from the returned roots / witnesses / betas afterwards. cap_height 0 only.
struct FriTailRound {
  uint32_t root[8];
  uint64_t witness;
  E2 beta;
};
struct FriTailRoll {
  const E2* p;
  uint32_t len;
  uint32_t pad;
};
// launches only: state_dev (8 words) is read and updated, records / final vector stay on the device
void fri_tail(Ctx& ctx, const E2* cur0, uint32_t len0, uint32_t n_rounds, unsigned pow_bits, uint32_t* state_dev,
              const std::vector<FriTailRoll>& rolls, Digest* tree_out, E2* layers_out, FriTailRound* rounds_dev, E2* final_dev);
// where one commit-phase round's challenger step reads its state and leaves its record (device pointers)
struct FriChallenge {
  uint32_t* state;
  FriTailRound* rec;
  uint32_t pow_bits;
};
// One commit-phase round in ONE launch (hash.hip::subtree_k): fold with the challenge `prev` left on the device, leaf
// digests of the folded layer, its whole tree and the challenger step `fc`. fusable: rows / 2 leaves fit the kernel.
bool fri_round_fusable(size_t rows);
void fri_round_fused(Ctx& ctx, DTree& t, const E2* cur, size_t rows, const FriTailRound* prev, const E2* roll_in, E2* out,
                     const FriChallenge& fc);
// fold with beta read from rec (device); next_leaves != nullptr also writes the next layer's leaf digests
// row0 / rows_total: `cur`, `roll_in`, `out` are the slice [row0, row0 + rows) of a folded layer of rows_total rows (0 = whole layer)
// squarings: the fold uses beta^(2^squarings) - step j of a round of arity 2^a, which is a binary folds with beta, beta^2, beta^4 ..
void fri_fold_dev(Ctx& ctx, const E2* cur, size_t rows, const FriTailRound* rec, const E2* roll_in /*nullable*/, E2* out,
                  Digest* next_leaves /*nullable*/, size_t row0 = 0, size_t rows_total = 0, unsigned squarings = 0);
// A FRI round whose tree is spread over ranks (prover_sharded.inc): `layer` holds the `len` sub-tree roots (2 <= len <= 1024, a
// power of two) with room for len - 1 more digests behind them; ONE launch hashes the levels above them into that room and runs
// the round's challenger step (observe the root, grind, sample beta: challenge_dev.h), as the last workgroup of subtree_k does
void merkle_top_challenge(Ctx& ctx, Digest* layer, size_t len, const FriChallenge& fc);
// cap of a tree + (when it fits) the PoW witness for transcript prefix || cap, in one host synchronisation
std::vector<Digest> cap_and_grind(Ctx& ctx, const DTree& t, const std::vector<uint8_t>& prefix, unsigned bits, bool* found,
                                  u64* witness);

}  // namespace msamd
