// logUp stage-2 trace construction and the claims accumulator on device.
// Replaces LookupValues::stage_2_traces (/root/reference/src/lookup.rs:472-555: messages, batch inverse, the
// serial running sum) and the serial claims loop of src/prover.rs:382-387. All sums are exact field sums, so a
// parallel scan / tree reduction gives bit-identical values to the reference's serial loops.
#include "lookup_params.h"
#include "msamd.h"
#include "program.h"

namespace msamd {

namespace {

constexpr int INV_CHUNK = 16;   // messages inverted together per thread (one base-field inversion, see e2_batch_inverse)
constexpr int CLAIMS_CHUNK = 16;
// m = beta + sum_i args[i] gamma^i (src/lookup.rs:375-384). With the powers precomputed every term is a
// base x ext product accumulated unreduced; otherwise Horner over the reversed args.
__device__ __forceinline__ E2 message(const u64* __restrict__ a, u32 n, E2 beta, E2 gamma, const GammaPows& gp) {
  if (n <= gp.n) {
    GlAcc a0, a1;
    acc_init(a0);
    acc_init(a1);
    for (u32 k = 0; k < n; k++) {
      const u64 v = a[k];
      acc_mad(a0, v, gp.g[k].c0);
      acc_mad(a1, v, gp.g[k].c1);
    }
    return e2(gl_add(acc_reduce(a0), beta.c0), gl_add(acc_reduce(a1), beta.c1));
  }
  E2 f = e2(0);
  for (u32 k = n; k-- > 0;) {
    f = e2_mul(f, gamma);
    f.c0 = gl_add(f.c0, a[k]);
  }
  return e2_add(f, beta);
}

// Visit every lookup j of one row with its inverse message; F(j, inv_msg).
template <class F>
__device__ __forceinline__ void for_each_inverse(const u64* __restrict__ args_row, const u32* __restrict__ offs, u32 L, E2 beta,
                                                 E2 gamma, const GammaPows& gp, F&& f) {
  for (u32 j0 = 0; j0 < L; j0 += INV_CHUNK) {
    E2 msg[INV_CHUNK];
    const int cnt = (int)(L - j0 < (u32)INV_CHUNK ? L - j0 : (u32)INV_CHUNK);
#pragma unroll
    for (int t = 0; t < INV_CHUNK; t++) {
      u32 j = j0 + t;
      if (j < L) msg[t] = message(args_row + offs[j], offs[j + 1] - offs[j], beta, gamma, gp);
    }
    e2_batch_inverse<INV_CHUNK>(msg, cnt);
#pragma unroll
    for (int t = 0; t < INV_CHUNK; t++) {
      u32 j = j0 + t;
      if (j < L) f(j, msg[t]);
    }
  }
}

// pass 1 (natural row order): terms[r][j] = mult * msg^-1 and the row total
__global__ __launch_bounds__(256) void stage2_terms_k(const u64* __restrict__ mult, const u64* __restrict__ args,
                                                      const u32* __restrict__ offs, size_t n, u32 L, u32 aw,
                                                      const ChallengeBG* __restrict__ ch, E2* __restrict__ terms, E2* __restrict__ rowsum) {
  size_t r = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (r >= n) return;
  const E2 beta = ch->beta, gamma = ch->gamma;
  const GammaPows& gp = ch->gp;
  E2 s = e2(0);
  const u64* mrow = mult + r * L;
  E2* trow = terms + r * L;
  for_each_inverse(args + r * aw, offs, L, beta, gamma, gp, [&](u32 j, E2 inv) {
    E2 t = e2_mul_base(inv, mrow[j]);
    trow[j] = t;
    s = e2_add(s, t);
  });
  rowsum[r] = s;
}

// pass 2: thread t owns storage row t = natural row bitrev(t): one scattered 16 L-byte row read, then 2L coalesced
// column stores of the running sum (the other way round costs 2L scattered 8-byte stores per row)
// (block_prefix: the scan's third step - adding each block's offset to its rows' prefixes - done here instead of in a pass of
// its own; rowprefix then holds prefixes inside blocks of `per` rows)
__global__ __launch_bounds__(256) void stage2_write_k(const E2* __restrict__ terms, size_t n, unsigned logn, u32 L,
                                                      const E2* __restrict__ rowprefix, u64* __restrict__ out,
                                                      const E2* __restrict__ block_prefix, u32 per) {
  size_t rr = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (rr >= n) return;
  const size_t r = bitrev64(rr, logn);
  E2 run = rowprefix[r];
  if (block_prefix) run = e2_add(run, block_prefix[r / per]);
  const E2* __restrict__ trow = terms + r * L;
  // eight terms in flight per thread: the row sits at a scattered address, so its loads must not wait for the sums
  for (u32 j0 = 0; j0 < L; j0 += 8) {
    E2 tv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) tv[k] = j0 + k < L ? trow[j0 + k] : e2(0);
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const u32 j = j0 + k;
      if (j < L) {
        out[size_t(2 * j) * n + rr] = run.c0;
        out[size_t(2 * j + 1) * n + rr] = run.c1;
        run = e2_add(run, tv[k]);
      }
    }
  }
}

// ---- exclusive scan of Ext2 values (field addition), three launches
constexpr int SCAN_ITEMS = 4;
__global__ __launch_bounds__(256) void scan_block_k(const E2* __restrict__ in, E2* __restrict__ out, size_t n, E2* __restrict__ block_tot) {
  __shared__ E2 sh[256];
  size_t base = (blockIdx.x * size_t(256) + threadIdx.x) * SCAN_ITEMS;
  E2 v[SCAN_ITEMS];
  E2 t = e2(0);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    v[k] = base + k < n ? in[base + k] : e2(0);
    t = e2_add(t, v[k]);
  }
  sh[threadIdx.x] = t;
  __syncthreads();
  for (u32 d = 1; d < 256; d <<= 1) {
    E2 x = threadIdx.x >= d ? sh[threadIdx.x - d] : e2(0);
    __syncthreads();
    sh[threadIdx.x] = e2_add(sh[threadIdx.x], x);
    __syncthreads();
  }
  E2 excl = threadIdx.x ? sh[threadIdx.x - 1] : e2(0);
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    if (base + k < n) out[base + k] = excl;
    excl = e2_add(excl, v[k]);
  }
  if (threadIdx.x == 255) block_tot[blockIdx.x] = sh[255];
}
// serial-over-tiles scan of the block totals by one workgroup; writes exclusive prefixes in place, total to *total
__global__ __launch_bounds__(256) void scan_totals_k(E2* __restrict__ tot, size_t nb, E2* __restrict__ total) {
  __shared__ E2 sh[256];
  __shared__ E2 carry;
  if (threadIdx.x == 0) carry = e2(0);
  __syncthreads();
  for (size_t base = 0; base < nb; base += 256) {
    size_t i = base + threadIdx.x;
    E2 v = i < nb ? tot[i] : e2(0);
    sh[threadIdx.x] = v;
    __syncthreads();
    for (u32 d = 1; d < 256; d <<= 1) {
      E2 x = threadIdx.x >= d ? sh[threadIdx.x - d] : e2(0);
      __syncthreads();
      sh[threadIdx.x] = e2_add(sh[threadIdx.x], x);
      __syncthreads();
    }
    E2 c = carry;
    if (i < nb) tot[i] = e2_add(c, e2_sub(sh[threadIdx.x], v));
    __syncthreads();
    if (threadIdx.x == 255) carry = e2_add(c, sh[255]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}
__global__ __launch_bounds__(256) void scan_add_k(E2* __restrict__ out, size_t n, const E2* __restrict__ block_prefix) {
  size_t base = (blockIdx.x * size_t(256) + threadIdx.x) * SCAN_ITEMS;
  E2 p = block_prefix[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++)
    if (base + k < n) out[base + k] = e2_add(out[base + k], p);
}

__global__ __launch_bounds__(256) void claims_acc_k(const u64* __restrict__ data, const u64* __restrict__ offs, size_t n,
                                                    const ChallengeBG* __restrict__ ch, E2* __restrict__ partial) {
  __shared__ E2 sh[256];
  const E2 beta = ch->beta, gamma = ch->gamma;
  const GammaPows& gp = ch->gp;
  const size_t base = blockIdx.x * size_t(256 * CLAIMS_CHUNK) + threadIdx.x;  // claim t of this thread: base + t * 256
  E2 msg[CLAIMS_CHUNK];
  E2 sum = e2(0);
  int cnt = 0;
#pragma unroll
  for (int t = 0; t < CLAIMS_CHUNK; t++) {
    size_t i = base + size_t(t) * 256;
    if (i < n) {
      msg[t] = message(data + offs[i], (u32)(offs[i + 1] - offs[i]), beta, gamma, gp);
      cnt = t + 1;
    }
  }
  if (cnt) {
    e2_batch_inverse<CLAIMS_CHUNK>(msg, cnt);
#pragma unroll
    for (int t = 0; t < CLAIMS_CHUNK; t++)
      if (t < cnt) sum = e2_add(sum, msg[t]);
  }
  sh[threadIdx.x] = sum;
  __syncthreads();
  for (u32 d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d) sh[threadIdx.x] = e2_add(sh[threadIdx.x], sh[threadIdx.x + d]);
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}

// words[0] = total; claim j: words[1 + j + offs[j]] = len_j, then its elements. The call covers claims [first, first + n):
// offs points at the offset of claim `first`, offsets are absolute, `data` and `words` are addressed absolutely (a rank
// that holds a slice passes pointers shifted back by the slice's start)
__global__ __launch_bounds__(256) void claims_words_k(const u64* __restrict__ data, const u64* __restrict__ offs, size_t n,
                                                      u64* __restrict__ words, size_t first, size_t total) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (i == 0 && first == 0) words[0] = (u64)total;
  if (i >= n) return;
  u64 o = offs[i], len = offs[i + 1] - o;
  u64* w = words + 1 + first + i + o;
  w[0] = len;
  for (u64 k = 0; k < len; k++) w[1 + k] = data[o + k];
}

}  // namespace

// exclusive prefix sums of in[0..n) into out; the grand total goes to *total_dev (device), nothing is read back
// (block_prefix_out: leave the last step - adding the blocks' offsets - to the consumer, which gets the offsets here)
constexpr u32 SCAN_PER_BLOCK = 256 * SCAN_ITEMS;
static void scan_exclusive(Ctx& ctx, const E2* in, E2* out, size_t n, E2* total_dev, DBuf<E2>* block_prefix_out = nullptr) {
  size_t per = SCAN_PER_BLOCK;
  size_t nb = (n + per - 1) / per;
  DBuf<E2> tot(ctx, nb);
  hipLaunchKernelGGL(scan_block_k, dim3((unsigned)nb), dim3(256), 0, ctx.stream, in, out, n, tot.p);
  hipLaunchKernelGGL(scan_totals_k, dim3(1), dim3(256), 0, ctx.stream, tot.p, nb, total_dev);
  if (block_prefix_out)
    *block_prefix_out = std::move(tot);
  else
    hipLaunchKernelGGL(scan_add_k, dim3((unsigned)nb), dim3(256), 0, ctx.stream, out, n, tot.p);
  HIP_CHECK(hipGetLastError());
}

// beta / gamma as a block in device memory (the form the kernels read); the staging copy is taken at once, so the host block
// may go out of scope
DBuf<ChallengeBG> challenge_bg_upload(Ctx& ctx, E2 beta, E2 gamma) {
  ChallengeBG h;
  challenge_bg_fill(h, beta, gamma);
  DBuf<ChallengeBG> d(ctx, 1);
  ctx.h2d(d.p, &h, sizeof(h));
  return d;
}

void stage2_build_async(Ctx& ctx, const DLookups& lk, E2 beta, E2 gamma, u64* out, E2* total_dev, const JitKernel* jit) {
  DBuf<ChallengeBG> ch = challenge_bg_upload(ctx, beta, gamma);
  stage2_build_dyn(ctx, lk, ch.p, out, total_dev, jit);
}
void stage2_build_dyn(Ctx& ctx, const DLookups& lk, const ChallengeBG* ch, u64* out, E2* total_dev, const JitKernel* jit) {
  size_t n = lk.height;
  if (lk.num_lookups == 0) {
    // pass-through accumulator column: zeros (src/lookup.rs:517-521)
    HIP_CHECK(hipMemsetAsync(out, 0, n * 2 * sizeof(u64), ctx.stream));
    HIP_CHECK(hipMemsetAsync(total_dev, 0, sizeof(E2), ctx.stream));
    return;
  }
  unsigned logn = log2_strict(n);
  u32 L = (u32)lk.num_lookups, aw = (u32)lk.args_width;
  DBuf<E2> rowsum(ctx, n), prefix(ctx, n), terms(ctx, n * L);
  dim3 grid((unsigned)((n + 255) / 256));
  hipEvent_t ev = ctx.prof_begin(K_STAGE2);
  if (jit && jit->function) {  // the circuit's own kernel: argument offsets are literals, the row is loaded up front
    Stage2Params sp;
    sp.mult = lk.mult.p;
    sp.args = lk.args.p;
    sp.n = n;
    sp.ch = ch;
    sp.terms = terms.p;
    sp.rowsum = rowsum.p;
    stage2_jit_launch(ctx, *jit, sp);
  } else {
    hipLaunchKernelGGL(stage2_terms_k, grid, dim3(256), 0, ctx.stream, lk.mult.p, lk.args.p, lk.arg_offsets.p, n, L, aw, ch, terms.p,
                       rowsum.p);
  }
  ctx.prof_end(K_STAGE2, ev, double(n) * 8.0 * (L + aw));
  DBuf<E2> block_prefix;
  scan_exclusive(ctx, rowsum.p, prefix.p, n, total_dev, &block_prefix);
  ev = ctx.prof_begin(K_STAGE2);
  hipLaunchKernelGGL(stage2_write_k, grid, dim3(256), 0, ctx.stream, (const E2*)terms.p, n, logn, L, (const E2*)prefix.p, out,
                     (const E2*)block_prefix.p, SCAN_PER_BLOCK);
  ctx.prof_end(K_STAGE2, ev, double(n) * 16.0 * L);
  HIP_CHECK(hipGetLastError());
}

void stage2_from_trace_async(Ctx& ctx, const JitKernel& trace_jit, const u64* d_trace, const u64* d_pre, size_t n, size_t num_lookups,
                             size_t args_width, E2 beta, E2 gamma, u64* out, E2* total_dev) {
  DBuf<ChallengeBG> ch = challenge_bg_upload(ctx, beta, gamma);
  stage2_from_trace_dyn(ctx, trace_jit, d_trace, d_pre, n, num_lookups, args_width, ch.p, out, total_dev);
}
void stage2_from_trace_dyn(Ctx& ctx, const JitKernel& trace_jit, const u64* d_trace, const u64* d_pre, size_t n, size_t num_lookups,
                           size_t args_width, const ChallengeBG* ch, u64* out, E2* total_dev) {
  if (!trace_jit.function || num_lookups == 0) throw std::runtime_error("stage2_from_trace: no fused kernel for this circuit");
  const unsigned logn = log2_strict(n);
  const u32 L = (u32)num_lookups;
  DBuf<E2> rowsum(ctx, n), prefix(ctx, n), terms(ctx, n * L);
  dim3 grid((unsigned)((n + 255) / 256));
  Stage2TraceParams sp;
  sp.trace = d_trace;
  sp.pre = d_pre;
  sp.n = n;
  sp.ch = ch;
  sp.terms = terms.p;
  sp.rowsum = rowsum.p;
  hipEvent_t ev = ctx.prof_begin(K_STAGE2);
  stage2_trace_jit_launch(ctx, trace_jit, sp);
  ctx.prof_end(K_STAGE2, ev, double(n) * 8.0 * (L + args_width));
  DBuf<E2> block_prefix;
  scan_exclusive(ctx, rowsum.p, prefix.p, n, total_dev, &block_prefix);
  ev = ctx.prof_begin(K_STAGE2);
  hipLaunchKernelGGL(stage2_write_k, grid, dim3(256), 0, ctx.stream, (const E2*)terms.p, n, logn, L, (const E2*)prefix.p, out,
                     (const E2*)block_prefix.p, SCAN_PER_BLOCK);
  ctx.prof_end(K_STAGE2, ev, double(n) * 16.0 * L);
  HIP_CHECK(hipGetLastError());
}

E2 stage2_build(Ctx& ctx, const DLookups& lk, E2 beta, E2 gamma, u64* out, const JitKernel* jit) {
  DBuf<E2> tot(ctx, 1);
  stage2_build_async(ctx, lk, beta, gamma, out, tot.p, jit);
  E2 total;
  ctx.d2h(&total, tot.p, sizeof(E2));
  return total;
}

void claims_accumulator_async(Ctx& ctx, const u64* d_data, const u64* d_offs, size_t n, E2 beta, E2 gamma, E2* out_dev) {
  DBuf<ChallengeBG> ch = challenge_bg_upload(ctx, beta, gamma);
  claims_accumulator_dyn(ctx, d_data, d_offs, n, ch.p, out_dev);
}
void claims_accumulator_dyn(Ctx& ctx, const u64* d_data, const u64* d_offs, size_t n, const ChallengeBG* ch, E2* out_dev) {
  if (n == 0) {
    HIP_CHECK(hipMemsetAsync(out_dev, 0, sizeof(E2), ctx.stream));
    return;
  }
  size_t per = 256 * CLAIMS_CHUNK;
  size_t nb = (n + per - 1) / per;
  DBuf<E2> partial(ctx, nb);
  hipLaunchKernelGGL(claims_acc_k, dim3((unsigned)nb), dim3(256), 0, ctx.stream, d_data, d_offs, n, ch, partial.p);
  hipLaunchKernelGGL(scan_totals_k, dim3(1), dim3(256), 0, ctx.stream, partial.p, nb, out_dev);  // only the total is used
  HIP_CHECK(hipGetLastError());
}

E2 claims_accumulator(Ctx& ctx, const u64* d_data, const u64* d_offs, size_t n, E2 beta, E2 gamma) {
  DBuf<E2> out(ctx, 1);
  claims_accumulator_async(ctx, d_data, d_offs, n, beta, gamma, out.p);
  E2 s;
  ctx.d2h(&s, out.p, sizeof(E2));
  return s;
}

size_t claims_transcript_words(Ctx& ctx, const u64* d_data, const u64* d_offs, size_t n, size_t total_elems, u64* d_words) {
  size_t nwords = 1 + n + total_elems;
  hipLaunchKernelGGL(claims_words_k, dim3((unsigned)((n + 255) / 256 + 1)), dim3(256), 0, ctx.stream, d_data, d_offs, n, d_words, size_t(0), n);
  HIP_CHECK(hipGetLastError());
  return nwords;
}

void claims_transcript_words_slice(Ctx& ctx, const u64* d_data_abs, const u64* d_offs_first, size_t first, size_t count, size_t n_total,
                                   u64* d_words_abs) {
  hipLaunchKernelGGL(claims_words_k, dim3((unsigned)((count + 255) / 256 + 1)), dim3(256), 0, ctx.stream, d_data_abs, d_offs_first, count,
                     d_words_abs, first, n_total);
  HIP_CHECK(hipGetLastError());
}

}  // namespace msamd
