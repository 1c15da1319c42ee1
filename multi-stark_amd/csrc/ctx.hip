// Device context: stream, twiddle tables, pooled memory, event-based per-kernel timing.
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <set>

#include "msamd.h"

namespace msamd {

// Read-backs queued by d2h_queue point at the caller's locals. When an error unwinds the call before the next
// synchronisation delivers them, those destinations are gone: the C-ABI catch blocks call abandon_pending() so that the
// entries are dropped (after a best-effort synchronisation) instead of being written through later.
namespace {
thread_local Ctx* tl_pending_ctx = nullptr;
std::mutex g_live_mu;
std::set<Ctx*> g_live;
}  // namespace
namespace {
struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  RoctxApi() {
    if (getenv("MSAMD_NO_ROCTX")) return;
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
      pop = (int (*)())dlsym(h, "roctxRangePop");
      if (push && pop) return;
      push = nullptr;
      pop = nullptr;
    }
  }
};
const RoctxApi& roctx() {
  static RoctxApi api;
  return api;
}
}  // namespace
RoctxRange::RoctxRange(const char* name) {
  if (roctx().push) {
    roctx().push(name);
    on = true;
  }
}
RoctxRange::~RoctxRange() {
  if (on) roctx().pop();
}
void RoctxRange::next(const char* name) {
  if (on) {
    roctx().pop();
    roctx().push(name);
  }
}

// MSAMD_ABORT_TRACE=1 (diagnostics): a native backtrace on SIGABRT - the runtime's own assertions and std::terminate say
// nothing about where they were raised
namespace {
struct sigaction g_prev_abort;  // the handler the host had installed (Python's faulthandler, a Rust panic hook): chained to
void abort_trace(int sig) {
  void* frames[64];
  const int n = backtrace(frames, 64);  // (the unwinder was loaded by install_abort_trace: no dlopen / malloc in here)
  static const char msg[] = "[msamd] SIGABRT, native backtrace of the aborting thread:\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  backtrace_symbols_fd(frames, n, 2);
  // hand the signal on: the host's own handler if it had one, the default action (core, exit status) otherwise
  if (g_prev_abort.sa_handler != SIG_DFL && g_prev_abort.sa_handler != SIG_IGN && g_prev_abort.sa_handler != abort_trace) {
    sigaction(SIGABRT, &g_prev_abort, nullptr);
  } else {
    signal(SIGABRT, SIG_DFL);
  }
  raise(sig);
}
void install_abort_trace() {
  static const bool once = [] {
    if (getenv("MSAMD_ABORT_TRACE")) {
      void* warm[4];
      (void)backtrace(warm, 4);  // the first call loads libgcc_s and allocates: not something to do inside a signal handler
      struct sigaction sa;
      memset(&sa, 0, sizeof(sa));
      sa.sa_handler = abort_trace;
      sigemptyset(&sa.sa_mask);
      sigaction(SIGABRT, &sa, &g_prev_abort);
    }
    return true;
  }();
  (void)once;
}
}  // namespace

namespace {
std::mutex g_pin_mu;  // one register / unregister at a time process-wide (thread ranks hand in buffers that share pages)
struct PinnedRange {
  size_t bytes;
  int refs;
};
std::map<const void*, PinnedRange> g_pinned_ranges;
}  // namespace
int host_range_pin(const void* p, size_t bytes) {
  if (!p || !bytes) return 1;
  std::lock_guard<std::mutex> lk(g_pin_mu);
  auto it = g_pinned_ranges.find(p);
  if (it != g_pinned_ranges.end() && it->second.bytes >= bytes) {
    it->second.refs++;
    return 1;
  }
  if (it != g_pinned_ranges.end()) return 0;  // the same start, now longer: the tail is not locked, and the earlier owners' lock stays as it is
  const hipError_t e = hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault);
  if (e == hipSuccess) {
    g_pinned_ranges[p] = PinnedRange{bytes, 1};
    return 1;
  }
  (void)hipGetLastError();
  return e == hipErrorHostMemoryAlreadyRegistered ? 2 : 0;
}
void host_range_unpin(const void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> lk(g_pin_mu);
  auto it = g_pinned_ranges.find(p);
  if (it == g_pinned_ranges.end()) return;
  if (--it->second.refs > 0) return;
  g_pinned_ranges.erase(it);
  (void)hipHostUnregister(const_cast<void*>(p));
  (void)hipGetLastError();  // (a range the caller has already freed: nothing to report)
}

void abandon_pending() {
  Ctx* c = tl_pending_ctx;
  tl_pending_ctx = nullptr;
  if (!c) return;
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    if (!g_live.count(c)) return;
  }
  (void)hipStreamSynchronize(c->copy_stream);
  (void)hipStreamSynchronize(c->claims_stream);
  c->side_join();
  (void)hipStreamSynchronize(c->main_stream);
  (void)hipGetLastError();
  c->down_pending.clear();
  c->down_used = 0;
  c->up_used = 0;
}

static const char* KNAMES[K_COUNT] = {"ntt_lds_strided", "ntt_lds_contig", "ntt12_dif", "ntt12_dit", "ntt8s_dif", "ntt8s_dit", "leaf_hash", "compress_layer", "stage2", "quotient",
                                      "bary_eval",   "deep_reduce", "fri_fold", "transpose",      "other"};
const char* kernel_name(int id) { return id >= 0 && id < K_COUNT ? KNAMES[id] : "?"; }

namespace {
__global__ void field_op_k(int op, const u64* a, const u64* b, size_t n, u64* out) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  switch (op) {
    case 0: out[i] = gl_add(a[i], b[i]); break;
    case 1: out[i] = gl_sub(a[i], b[i]); break;
    case 2: out[i] = gl_mul(a[i], b[i]); break;
    case 3: out[i] = gl_inv(a[i]); break;
    case 4: {
      E2 r = e2_mul(e2(a[2 * i], a[2 * i + 1]), e2(b[2 * i], b[2 * i + 1]));
      out[2 * i] = r.c0;
      out[2 * i + 1] = r.c1;
      break;
    }
    default: {
      E2 r = e2_inv(e2(a[2 * i], a[2 * i + 1]));
      out[2 * i] = r.c0;
      out[2 * i + 1] = r.c1;
      break;
    }
  }
}
}  // namespace

void field_op(Ctx& ctx, int op, const u64* a, const u64* b, size_t n, u64* out) {
  hipLaunchKernelGGL(field_op_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx.stream, op, a, b, n, out);
  HIP_CHECK(hipGetLastError());
}

Ctx::Ctx(int dev) : device(dev) {
  install_abort_trace();
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live.insert(this);
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) throw std::runtime_error("no HIP device available");
  if (dev < 0 || dev >= count) throw std::runtime_error("HIP device index out of range");
  HIP_CHECK(hipSetDevice(dev));
  HIP_CHECK(hipStreamCreate(&stream));
  main_stream = stream;
  HIP_CHECK(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
  for (auto& e : side_ev) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  side_config();
  HIP_CHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&claims_stream, hipStreamNonBlocking));
  for (auto& e : copy_ev) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  pinned_half = size_t(8) << 20;
  if (hipHostMalloc((void**)&pinned, 2 * pinned_half + 256, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    pinned = nullptr;  // transfers fall back to pageable staging by the runtime
  }
  if (hipHostMalloc((void**)&bounce, BOUNCE_BYTES, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    bounce = nullptr;  // (synchronous hipMemcpy instead)
  }
  if (pinned) {
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, pinned, 0) == hipSuccess && dp) {
      pinned_dev = (uint8_t*)dp;
      flag_host = reinterpret_cast<uint32_t*>(pinned + 2 * pinned_half);
      flag_dev = reinterpret_cast<uint32_t*>(pinned_dev + 2 * pinned_half);
      *flag_host = 0;
    } else {
      (void)hipGetLastError();
    }
  }
  const size_t T = size_t(1) << TW_HALF;
  std::vector<u64> h(4 * T);
  u64 W = gl_two_adic_generator(TW_LOG), Wi = gl_inv(W);
  u64 Wh = gl_exp_pow2(W, TW_HALF), Whi = gl_exp_pow2(Wi, TW_HALF);
  u64 a = 1, b = 1, c = 1, d = 1;
  for (size_t i = 0; i < T; i++) {
    h[i] = a;
    h[T + i] = b;
    h[2 * T + i] = c;
    h[3 * T + i] = d;
    a = gl_mul(a, W);
    b = gl_mul(b, Wh);
    c = gl_mul(c, Wi);
    d = gl_mul(d, Whi);
  }
  u64* t = nullptr;
  HIP_CHECK(hipMalloc(&t, 4 * T * sizeof(u64)));
  HIP_CHECK(hipMemcpy(t, h.data(), 4 * T * sizeof(u64), hipMemcpyHostToDevice));
  tw0 = t;
  tw1 = t + T;
  tw0i = t + 2 * T;
  tw1i = t + 3 * T;
  // compact tables for roots of order 2^1 .. 2^12
  std::vector<u64> cf(4096, 0), ci(4096, 0);
  for (unsigned r = 1; r <= 12; r++) {
    u64 w = gl_two_adic_generator(r), wi = gl_inv(w), x = 1, y = 1;
    size_t off = (size_t(1) << (r - 1)) - 1;
    for (size_t i = 0; i < (size_t(1) << (r - 1)); i++) {
      cf[off + i] = x;
      ci[off + i] = y;
      x = gl_mul(x, w);
      y = gl_mul(y, wi);
    }
  }
  std::vector<u64> c3f(2048, 0), c3i(2048, 0);
  for (unsigned r = 2; r <= 12; r++) {
    u64 w = gl_two_adic_generator(r), wi = gl_inv(w);
    u64 w3 = gl_mul(w, gl_mul(w, w)), wi3 = gl_mul(wi, gl_mul(wi, wi)), x = 1, y = 1;
    size_t off = (size_t(1) << (r - 2)) - 1;
    for (size_t i = 0; i < (size_t(1) << (r - 2)); i++) {
      c3f[off + i] = x;
      c3i[off + i] = y;
      x = gl_mul(x, w3);
      y = gl_mul(y, wi3);
    }
  }
  u64* tc = nullptr;
  HIP_CHECK(hipMalloc(&tc, (2 * 4096 + 2 * 2048) * sizeof(u64)));
  HIP_CHECK(hipMemcpy(tc, cf.data(), 4096 * sizeof(u64), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(tc + 4096, ci.data(), 4096 * sizeof(u64), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(tc + 8192, c3f.data(), 2048 * sizeof(u64), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(tc + 8192 + 2048, c3i.data(), 2048 * sizeof(u64), hipMemcpyHostToDevice));
  twc = tc;
  twci = tc + 4096;
  twc3 = tc + 8192;
  twc3i = tc + 8192 + 2048;
  std::vector<u64> ff(8192, 0), fi(8192, 0);
  for (unsigned r = 1; r <= 12; r++) {
    u64 w = gl_two_adic_generator(r), wi = gl_inv(w), x = 1, y = 1;
    size_t off = (size_t(1) << r) - 1;
    for (size_t i = 0; i < (size_t(1) << r); i++) {
      ff[off + i] = x;
      fi[off + i] = y;
      x = gl_mul(x, w);
      y = gl_mul(y, wi);
    }
  }
  u64* tf = nullptr;
  HIP_CHECK(hipMalloc(&tf, 2 * 8192 * sizeof(u64)));
  HIP_CHECK(hipMemcpy(tf, ff.data(), 8192 * sizeof(u64), hipMemcpyHostToDevice));
  HIP_CHECK(hipMemcpy(tf + 8192, fi.data(), 8192 * sizeof(u64), hipMemcpyHostToDevice));
  twf = tf;
  twfi = tf + 8192;
  HIP_CHECK(hipMalloc(&tree_counter, 256));
  HIP_CHECK(hipMemset(tree_counter, 0, 256));
}

Ctx::~Ctx() {
  {
    std::lock_guard<std::mutex> lk(g_live_mu);
    g_live.erase(this);
  }
  (void)hipSetDevice(device);
  if (copy_stream) (void)hipStreamSynchronize(copy_stream);
  if (claims_stream) (void)hipStreamSynchronize(claims_stream);
  if (side_stream) (void)hipStreamSynchronize(side_stream);
  stream = main_stream;
  (void)hipStreamSynchronize(stream);
  for (auto& p : prof_pending) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  for (auto e : event_pool) (void)hipEventDestroy(e);
  for (auto& kv : pool_free) (void)hipFree(kv.second);
  for (auto& kv : pool_live) (void)hipFree(kv.first);
  for (auto& kv : pool_free_side) (void)hipFree(kv.second);
  for (auto& kv : side_live) (void)hipFree(kv.first);
  for (auto& kv : side_deferred) (void)hipFree(kv.second);
  for (auto e : side_ev)
    if (e) (void)hipEventDestroy(e);
  if (side_stream) (void)hipStreamDestroy(side_stream);
  for (auto& kv : lde_scales) (void)hipFree(kv.second);
  if (pinned) (void)hipHostFree(pinned);
  if (bounce) (void)hipHostFree(bounce);
  if (tw0) (void)hipFree(tw0);
  if (twc) (void)hipFree(twc);
  if (twf) (void)hipFree(twf);
  if (tree_counter) (void)hipFree(tree_counter);
  for (auto e : copy_ev)
    if (e) (void)hipEventDestroy(e);
  if (copy_stream) (void)hipStreamDestroy(copy_stream);
  if (claims_stream) (void)hipStreamDestroy(claims_stream);
  for (auto e : group_events) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(stream);
}

void* Ctx::alloc(size_t bytes) {
  if (fail_alloc_countdown > 0 && --fail_alloc_countdown == 0) throw std::runtime_error("injected allocation failure (ms_ctx_debug_fail_alloc)");
  size_t sz = (bytes + 255) & ~size_t(255);
  if (sz == 0) sz = 256;
  const bool side = side_depth > 0;
  std::multimap<size_t, void*>& free_list = side ? pool_free_side : pool_free;
  auto it = free_list.find(sz);
  void* p = nullptr;
  if (it != free_list.end()) {
    p = it->second;
    free_list.erase(it);
  } else {
    hipError_t e = hipMalloc(&p, sz);
    if (e != hipSuccess) {
      // give pooled blocks back and retry once
      (void)hipGetLastError();
      trim();
      HIP_CHECK(hipMalloc(&p, sz));
    }
    pool_bytes += sz;
  }
  if (side)
    side_live[p] = sz;
  else
    pool_live[p] = sz;
  return p;
}

void Ctx::release(void* p) {
  auto it = pool_live.find(p);
  if (it != pool_live.end()) {
    pool_free.emplace(it->second, p);
    pool_live.erase(it);
    return;
  }
  it = side_live.find(p);
  if (it == side_live.end()) return;
  // a side block may still be in use by launches of the side stream that the main stream has not waited for
  if (side_forked)
    side_deferred.emplace_back(it->second, p);
  else
    pool_free_side.emplace(it->second, p);
  side_live.erase(it);
}

void Ctx::trim() {
  side_join();
  (void)hipStreamSynchronize(main_stream);
  for (auto& kv : pool_free) {
    (void)hipFree(kv.second);
    pool_bytes -= kv.first;
  }
  pool_free.clear();
  for (auto& kv : pool_free_side) {
    (void)hipFree(kv.second);
    pool_bytes -= kv.first;
  }
  pool_free_side.clear();
}

// MSAMD_SIDE_DELAY_US=n (diagnostics): every fork starts the side stream n microseconds late (one thread watching the
// constant 100 MHz clock; it ends by itself). The side stream normally finishes well before the main stream, which hides a
// missing join; with the delay a consumer on the main stream that does not wait for the side stream reads stale data every
// time (tests/test_blake3_circuit.py::test_side_stream_results_are_awaited). MSAMD_MAIN_DELAY_US=n delays the MAIN stream
// behind every fork instead: a side-stream launch that reads what the main stream was given after the fork then fails.
namespace {
__global__ void side_delay_k(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
}  // namespace

void Ctx::side_config() {
  side_enabled = !getenv("MSAMD_NO_SIDE_STREAM");
  const char* v = getenv("MSAMD_SIDE_MAX_LOG");
  side_max_log = v ? (unsigned)atoi(v) : 12u;
  if (side_max_log > 40) side_max_log = 40;
  const char* d = getenv("MSAMD_SIDE_DELAY_US");
  side_delay_us = d ? (unsigned)std::min(atoi(d), 20000) : 0u;
  d = getenv("MSAMD_MAIN_DELAY_US");
  main_delay_us = d ? (unsigned)std::min(atoi(d), 20000) : 0u;
  d = getenv("MSAMD_COPY_DELAY_US");
  copy_delay_us = d ? (unsigned)std::min(atoi(d), 20000) : 0u;
}

// MSAMD_COPY_DELAY_US=n (diagnostics): the copy stream starts n microseconds late at every proof's upload, so that a kernel
// which reads uploaded data (traces, lookup values, claims) without waiting for the upload's event reads stale data every time
void Ctx::copy_delay() {
  if (!copy_delay_us) return;
  hipLaunchKernelGGL(side_delay_k, dim3(1), dim3(1), 0, copy_stream, (unsigned long long)copy_delay_us * 100ull);
  HIP_CHECK(hipGetLastError());
}

void Ctx::side_fork() {
  if (!side_enabled) return;
  if (side_forked) side_join();
  HIP_CHECK(hipEventRecord(side_ev[0], main_stream));
  HIP_CHECK(hipStreamWaitEvent(side_stream, side_ev[0], 0));
  side_forked = true;
  if (side_delay_us) {
    hipLaunchKernelGGL(side_delay_k, dim3(1), dim3(1), 0, side_stream, (unsigned long long)side_delay_us * 100ull);
    HIP_CHECK(hipGetLastError());
  }
  if (main_delay_us) {  // the other direction: the side stream runs ahead of what the main stream is given AFTER the fork
    hipLaunchKernelGGL(side_delay_k, dim3(1), dim3(1), 0, main_stream, (unsigned long long)main_delay_us * 100ull);
    HIP_CHECK(hipGetLastError());
  }
}

void Ctx::side_join() {
  if (!side_forked) return;
  side_forked = false;
  hipError_t e = hipEventRecord(side_ev[1], side_stream);
  if (e == hipSuccess) e = hipStreamWaitEvent(main_stream, side_ev[1], 0);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    (void)hipStreamSynchronize(side_stream);  // fall back to a host-side wait: the blocks below must be idle
  }
  for (auto& kv : side_deferred) pool_free_side.emplace(kv.first, kv.second);
  side_deferred.clear();
}

// Small read-backs without the runtime's copy + wake-up path: ONE kernel moves every queued segment into the pinned
// (host-coherent) staging buffer and then raises a flag there with a system-scope release; the host polls the flag.
// A proof synchronises half a dozen times on a few hundred bytes (a cap, the logUp totals, the opened values), and the
// blocking wait's wake-up alone left the GPU idle for 35-50 us each time.
namespace {
struct FlagCopySeg {
  const uint8_t* src;
  uint32_t dst_off, n;  // byte offset in the staging half, byte count (multiples of 4 take the word path)
};
struct FlagCopyArgs {
  FlagCopySeg seg[24];
  uint32_t n_seg, seq;
  uint8_t* host_base;   // device-visible address of the staging half
  uint32_t* flag;       // device-visible address of the flag word
};
__global__ __launch_bounds__(256) void flag_copy_k(FlagCopyArgs a) {
  for (uint32_t s = 0; s < a.n_seg; s++) {
    const FlagCopySeg sg = a.seg[s];
    if (((sg.n | sg.dst_off) & 3u) == 0 && (reinterpret_cast<uintptr_t>(sg.src) & 3u) == 0) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(sg.src);
      uint32_t* dst = reinterpret_cast<uint32_t*>(a.host_base + sg.dst_off);
      for (uint32_t i = threadIdx.x; i < sg.n / 4; i += blockDim.x) dst[i] = src[i];
    } else {
      for (uint32_t i = threadIdx.x; i < sg.n; i += blockDim.x) a.host_base[sg.dst_off + i] = sg.src[i];
    }
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace

void Ctx::sync_and_deliver() {
  host_syncs++;
  side_join();  // read-backs and the staging halves are shared: everything the side stream was given completes first
  hipStream_t stream = main_stream;
  bool done = false;
  // Segments that were copied into the staging half when they were queued (long ones: a DMA copy on this stream) need nothing
  // but to have completed - the flag kernel runs behind them in the stream; the short ones are moved by the kernel itself.
  size_t n_short = 0;
  for (auto& d : down_pending) n_short += d.copied ? 0 : 1;
  if (flag_host && !down_pending.empty() && n_short <= 24 && !down_direct && !getenv("MSAMD_NO_FLAG_SYNC")) {
    FlagCopyArgs a;
    memset(&a, 0, sizeof(a));
    size_t total = 0, k = 0;
    for (size_t i = 0; i < down_pending.size(); i++) {
      if (down_pending[i].copied) continue;
      a.seg[k].src = (const uint8_t*)down_pending[i].src;
      a.seg[k].dst_off = (uint32_t)down_pending[i].off;
      a.seg[k].n = (uint32_t)down_pending[i].n;
      total += down_pending[i].n;
      k++;
    }
    if (total <= (size_t(64) << 10)) {
      a.n_seg = (uint32_t)k;
      a.seq = ++flag_seq;
      a.host_base = pinned_dev + pinned_half;
      a.flag = flag_dev;
      hipLaunchKernelGGL(flag_copy_k, dim3(1), dim3(256), 0, stream, a);
      HIP_CHECK(hipGetLastError());
      volatile uint32_t* f = flag_host;
      // poll; a stream that has failed never raises the flag: look at it now and then
      for (uint64_t spins = 0;; spins++) {
        if (*f == a.seq) {
          done = true;
          break;
        }
        if ((spins & 0xFFFFF) == 0xFFFFF) {
          hipError_t q = hipStreamQuery(stream);
          if (q != hipErrorNotReady) {
            HIP_CHECK(q);
            done = *f == a.seq;
            break;
          }
        }
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
  }
  if (!done) {
    for (auto& d : down_pending)
      if (!d.copied) HIP_CHECK(hipMemcpyAsync(pinned + pinned_half + d.off, d.src, d.n, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
  }
  for (auto& d : down_pending)
    if (d.dst) memcpy(d.dst, pinned + pinned_half + d.off, d.n);
  down_pending.clear();
  down_direct = false;
  down_used = 0;
  up_used = 0;  // every queued upload has executed
  if (side_depth > 0) side_fork();  // synchronised from inside a SideScope: what the scope queues next is joined again
}

void Ctx::bounce_h2d(void* dst, const void* src, size_t n, hipStream_t on) {
  hipStream_t s = on ? on : stream;
  if (!bounce) {
    HIP_CHECK(hipStreamSynchronize(s));
    HIP_CHECK(hipMemcpy(dst, src, n, hipMemcpyHostToDevice));
    return;
  }
  for (size_t off = 0; off < n; off += BOUNCE_BYTES) {
    const size_t m = std::min(BOUNCE_BYTES, n - off);
    memcpy(bounce, (const uint8_t*)src + off, m);
    HIP_CHECK(hipMemcpyAsync((uint8_t*)dst + off, bounce, m, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));  // (the buffer is reused by the next chunk; set-up paths only)
  }
}
void Ctx::bounce_d2h(void* dst, const void* src, size_t n) {
  if (!bounce) {
    HIP_CHECK(hipStreamSynchronize(stream));
    HIP_CHECK(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
    return;
  }
  for (size_t off = 0; off < n; off += BOUNCE_BYTES) {
    const size_t m = std::min(BOUNCE_BYTES, n - off);
    HIP_CHECK(hipMemcpyAsync(bounce, (const uint8_t*)src + off, m, hipMemcpyDeviceToHost, stream));
    HIP_CHECK(hipStreamSynchronize(stream));
    memcpy((uint8_t*)dst + off, bounce, m);
  }
}

void Ctx::h2d(void* dst, const void* src, size_t n) {
  if (n == 0) return;
  if (pinned && n <= (size_t(1) << 20)) {
    size_t need = (n + 63) & ~size_t(63);
    if (up_used + need > pinned_half) sync_and_deliver();
    memcpy(pinned + up_used, src, n);
    HIP_CHECK(hipMemcpyAsync(dst, pinned + up_used, n, hipMemcpyHostToDevice, stream));
    up_used += need;
    return;
  }
  bounce_h2d(dst, src, n);  // (complete on return)
}

// A read-back that is issued at once (long segments, pageable destinations) is a copy on the CURRENT stream: queued on the main
// stream while the side stream is still producing part of the source - the opened values of the short circuits beside the
// long ones', pcs_open - it would read those values before they exist (seen as wrong opened values of exactly the side-stream
// matrices in the FIRST proof of a wide system, when fresh allocations delay the side stream's launches; short segments are
// copied by sync_and_deliver, which joins first). The main stream therefore waits for the side stream here.
void Ctx::join_side_for_copy() {
  // join, then fork again at once: the session stays open, so the SideScopes that follow in the same phase keep the short
  // circuits on the side stream (ending the session here serialised them behind every direct read-back)
  if (side_forked && side_depth == 0) {
    side_join();
    side_fork();
  }
}

void Ctx::d2h_queue(void* dst, const void* src, size_t n) {
  if (n == 0) return;
  size_t need = (n + 63) & ~size_t(63);
  if (!pinned || need > pinned_half) {
    join_side_for_copy();
    bounce_d2h(dst, src, n);  // (delivered at once)
    return;
  }
  if (down_used + need > pinned_half) {
    // the staging half is full. Wrapping it (synchronise, deliver, start again at offset 0) would let the segments queued
    // next overwrite a view handed out by d2h_queue_staged that its caller has not read yet: with such a view pending, this
    // read-back goes to its destination at once instead, through the bounce buffer
    bool view_pending = false;
    for (auto& d : down_pending) view_pending = view_pending || d.dst == nullptr;
    if (view_pending) {
      join_side_for_copy();
      bounce_d2h(dst, src, n);  // (delivered at once)
      return;
    }
    sync_and_deliver();
  }
  // the copy itself is issued by the next synchronisation: one flag-copy kernel for all short segments, or one
  // hipMemcpyAsync each when they are many or long
  PendingD2H pd;
  pd.dst = dst;
  pd.src = src;
  pd.off = down_used;
  pd.n = n;
  pd.copied = false;
  if (!flag_host || n > (size_t(64) << 10)) {
    join_side_for_copy();
    HIP_CHECK(hipMemcpyAsync(pinned + pinned_half + down_used, src, n, hipMemcpyDeviceToHost, stream));
    pd.copied = true;  // (into the pinned staging half: complete once the stream has passed it - the flag kernel is enough)
    if (!flag_host) down_direct = true;
  }
  down_pending.push_back(pd);
  down_used += need;
  tl_pending_ctx = this;
}

const uint8_t* Ctx::d2h_queue_staged(const void* src, size_t n) {
  const size_t need = (n + 63) & ~size_t(63);
  if (!pinned || n == 0 || down_used + need > pinned_half) return nullptr;
  PendingD2H pd;
  pd.dst = nullptr;  // nothing to deliver: the caller reads the staging buffer
  pd.src = src;
  pd.off = down_used;
  pd.n = n;
  pd.copied = false;
  if (!flag_host || n > (size_t(64) << 10)) {
    join_side_for_copy();
    HIP_CHECK(hipMemcpyAsync(pinned + pinned_half + down_used, src, n, hipMemcpyDeviceToHost, stream));
    pd.copied = true;  // (into the pinned staging half: complete once the stream has passed it - the flag kernel is enough)
    if (!flag_host) down_direct = true;
  }
  down_pending.push_back(pd);
  const uint8_t* at = pinned + pinned_half + down_used;
  down_used += need;
  tl_pending_ctx = this;
  return at;
}

void Ctx::d2h(void* dst, const void* src, size_t n) {
  d2h_queue(dst, src, n);
  sync_and_deliver();
}

hipEvent_t Ctx::group_event(size_t i) {
  while (group_events.size() <= i) {
    hipEvent_t e;
    HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    group_events.push_back(e);
  }
  return group_events[i];
}

hipEvent_t Ctx::prof_begin(int id) {
  if (!prof_on(id)) return nullptr;
  hipEvent_t a;
  if (!event_pool.empty()) {
    a = event_pool.back();
    event_pool.pop_back();
  } else {
    HIP_CHECK(hipEventCreate(&a));
  }
  HIP_CHECK(hipEventRecord(a, stream));
  return a;
}

void Ctx::prof_end(int id, hipEvent_t a, double bytes) {
  if (!a) return;
  hipEvent_t b;
  if (!event_pool.empty()) {
    b = event_pool.back();
    event_pool.pop_back();
  } else {
    HIP_CHECK(hipEventCreate(&b));
  }
  HIP_CHECK(hipEventRecord(b, stream));
  prof_pending.push_back(Pending{id, a, b, bytes});
}

void Ctx::prof_collect() {
  if (prof_pending.empty()) return;
  side_join();
  HIP_CHECK(hipStreamSynchronize(main_stream));
  for (auto& p : prof_pending) {
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, p.a, p.b));
    stats[p.id].launches++;
    stats[p.id].ms += ms;
    stats[p.id].alg_bytes += p.bytes;
    event_pool.push_back(p.a);
    event_pool.push_back(p.b);
  }
  prof_pending.clear();
}

}  // namespace msamd
