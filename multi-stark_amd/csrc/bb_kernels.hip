// HIP kernels (gfx950) of the BabyBear / Poseidon2 path: transforms, Poseidon2 Merkle trees, logUp stage-2 traces,
// quotient evaluation, barycentric / DEEP / FRI-fold kernels. See bb.h for the data layout. Every kernel is integer
// VALU work on 32-bit Montgomery words; nothing here is shaped for MFMA.
#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "bb.h"
#include "bb_quotient_params.h"

namespace msbb {

using msamd::PNode;

static inline unsigned blocks_for(size_t n, unsigned t) { return (unsigned)((n + t - 1) / t); }
// HIP-event timing of one launch for a kernel class (ms_ctx_set_profile_mask): the classes are those of the Goldilocks
// path; `units` counts Poseidon2 permutations for the hash classes (bench.py prices them against the VALU issue peak)
struct ProfScope {
  Ctx& ctx;
  int id;
  hipEvent_t ev;
  double bytes, units;
  ProfScope(Ctx& c, int kid, double alg_bytes, double n_units = 0) : ctx(c), id(kid), ev(c.prof_begin(kid)), bytes(alg_bytes), units(n_units) {}
  ~ProfScope() {
    if (ev) ctx.stats[id].units += units;
    ctx.prof_end(id, ev, bytes);
  }
};

__device__ __forceinline__ size_t bitrev_dev(size_t x, unsigned bits) { return bits ? (size_t)(__brevll((unsigned long long)x) >> (64 - bits)) : 0; }

// ------------------------------------------------------------------ layout / conversion
__global__ void upload_rows_k(const u32* __restrict__ in, size_t h, size_t w, u32* __restrict__ out, size_t ld) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= h * w) return;
  size_t r = idx / w, c = idx % w;
  out[c * ld + r] = bb_to_monty(in[idx]);
}
__global__ void download_rows_k(const u32* __restrict__ in, size_t ld, size_t h, size_t w, unsigned log_h, int bitrev, u32* __restrict__ out) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= h * w) return;
  size_t r = idx / w, c = idx % w;
  size_t src = bitrev ? bitrev_dev(r, log_h) : r;
  out[idx] = bb_from_monty(in[c * ld + src]);
}
void bb_upload_rows(Ctx& ctx, const u32* host, size_t h, size_t w, BMat& out) {
  out = bmat(ctx, h, w);
  if (!h || !w) return;
  DBuf<u32> tmp(ctx, h * w);
  ctx.h2d(tmp.p, host, h * w * 4);
  upload_rows_k<<<blocks_for(h * w, 256), 256, 0, ctx.stream>>>(tmp.p, h, w, out.buf.p, out.ld);
  ctx.sync();
}
// the same without the final synchronisation, from (page-locked) caller memory: the copy and the conversion are queued on the
// context's stream; the staging block returns to the pool at once (the pool is stream-ordered)
void bb_upload_rows_async(Ctx& ctx, const u32* host, size_t h, size_t w, BMat& out) {
  out = bmat(ctx, h, w);
  if (!h || !w) return;
  DBuf<u32> tmp(ctx, h * w);
  HIP_CHECK(hipMemcpyAsync(tmp.p, host, h * w * 4, hipMemcpyHostToDevice, ctx.stream));
  upload_rows_k<<<blocks_for(h * w, 256), 256, 0, ctx.stream>>>(tmp.p, h, w, out.buf.p, out.ld);
  HIP_CHECK(hipGetLastError());
}
static unsigned log2_host(size_t n) {
  unsigned l = 0;
  while ((size_t(1) << l) < n) l++;
  return l;
}
void bb_download_rows(Ctx& ctx, const BMat& m, bool bitrev_rows, u32* host) {
  if (!m.h || !m.w) return;
  DBuf<u32> tmp(ctx, m.h * m.w);
  download_rows_k<<<blocks_for(m.h * m.w, 256), 256, 0, ctx.stream>>>(m.buf.p, m.ld, m.h, m.w, log2_host(m.h), bitrev_rows ? 1 : 0, tmp.p);
  ctx.d2h(host, tmp.p, m.h * m.w * 4);
}

// ------------------------------------------------------------------ transforms
// Twiddle table of the order-2^n root: w^j, j < 2^(n-1), Montgomery form. Cached per size in a process-wide map keyed
// by (device, n); the tables are small next to the matrices (half a column).
__global__ void twiddles_k(u32* out, size_t count, u32 w) {
  size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < count) out[j] = bb_pow(w, j);
}
struct TwCache {
  std::mutex mu;  // contexts of different devices may be driven from different host threads
  std::map<std::pair<int, unsigned>, u32*> t;
};
static TwCache& tw_cache() {
  static TwCache c;
  return c;
}
static const u32* twiddles(Ctx& ctx, unsigned log_n) {
  auto key = std::make_pair(ctx.device, log_n);
  std::lock_guard<std::mutex> lock(tw_cache().mu);
  auto it = tw_cache().t.find(key);
  if (it != tw_cache().t.end()) return it->second;
  size_t count = log_n ? (size_t(1) << (log_n - 1)) : 1;
  u32* p = nullptr;
  HIP_CHECK(hipMalloc(&p, count * 4));
  twiddles_k<<<blocks_for(count, 256), 256, 0, ctx.stream>>>(p, count, bb_two_adic_generator(log_n));
  tw_cache().t[key] = p;
  return p;
}

// DIF layers [l0, l0 + K) of a size-2^n transform; tile = 2^K values of the "middle" index bits x T consecutive low
// indices (128-byte runs), staged through LDS. Column = blockIdx.y.
template <int K, int T>
__global__ __launch_bounds__(256) void ntt_strided_k(u32* __restrict__ data, size_t ld, unsigned n, unsigned l0, const u32* __restrict__ tw) {
  __shared__ u32 s[(1 << K) * T];
  u32* col = data + (size_t)blockIdx.y * ld;
  const unsigned sbits = n - l0 - K;
  const size_t lo_tiles = (size_t(1) << sbits) / T;
  const size_t H = blockIdx.x / lo_tiles, lo0 = (blockIdx.x % lo_tiles) * T;
  const size_t base = (H << (n - l0)) + lo0;
  for (unsigned e = threadIdx.x; e < (1u << K) * T; e += 256) s[e] = col[base + ((size_t)(e / T) << sbits) + (e % T)];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < K; u++) {
    const int bit = K - 1 - u;
    for (unsigned b = threadIdx.x; b < (1u << (K - 1)) * T; b += 256) {
      unsigned t = b % T, m = b / T;
      unsigned mlo = m & ((1u << bit) - 1), mhi = m >> bit;
      unsigned m0 = (mhi << (bit + 1)) | mlo, m1 = m0 | (1u << bit);
      u32 x = s[m0 * T + t], y = s[m1 * T + t];
      size_t j = ((size_t)mlo << sbits) | (lo0 + t);
      u32 w = tw[j << (l0 + u)];
      s[m0 * T + t] = bb_add(x, y);
      s[m1 * T + t] = bb_mul(bb_sub(x, y), w);
    }
    __syncthreads();
  }
  for (unsigned e = threadIdx.x; e < (1u << K) * T; e += 256) col[base + ((size_t)(e / T) << sbits) + (e % T)] = s[e];
}
// Eight DIF layers [l0, l0 + 8) per pass with the data in registers: a tile is 256 values of the "middle" index bits x 16
// consecutive low indices; every thread runs two rounds of four layers on 16 values it holds (one LDS exchange between
// the rounds, padded to stay conflict-free). The twiddle of layer u of a round factors into a per-thread root A^(2^u)
// (one table load, then squarings) and a 16th root of unity that depends only on the register index (8 constants), so a
// round costs 46 multiplications per 16 values and no twiddle traffic.
__device__ __forceinline__ void r16_round(u32 (&x)[16], u32 A, const u32 (&C)[8]) {
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int half = 8 >> u;
    u32 T[8];
    T[0] = A;
#pragma unroll
    for (int jl = 1; jl < half; jl++) T[jl] = bb_mul(A, C[jl << u]);
#pragma unroll
    for (int blk = 0; blk < 16; blk += 2 * half)
#pragma unroll
      for (int jl = 0; jl < half; jl++) {
        u32 a = x[blk + jl], b = x[blk + jl + half];
        x[blk + jl] = bb_add(a, b);
        x[blk + jl + half] = bb_mul(bb_sub(a, b), T[jl]);
      }
    A = bb_mul(A, A);
  }
}
template <int LO>  // consecutive low indices per tile: 16 (64-byte runs) or 32 (128-byte runs)
__global__ __launch_bounds__(16 * LO) void ntt_r16_k(u32* __restrict__ data, size_t ld, unsigned n, unsigned l0, const u32* __restrict__ tw) {
  __shared__ u32 sh[256 * LO + 16 * LO];
  u32* col = data + (size_t)blockIdx.y * ld;
  const unsigned sbits = n - l0 - 8;
  const size_t lo_tiles = (size_t(1) << sbits) / LO;
  const size_t H = blockIdx.x / lo_tiles, lo0 = (blockIdx.x % lo_tiles) * LO;
  const size_t base = (H << (n - l0)) + lo0;
  const unsigned lo = threadIdx.x % LO, q = threadIdx.x / LO;
  u32 C[8];
#pragma unroll
  for (int k = 0; k < 8; k++) C[k] = tw[(size_t)k << (n - 4)];  // 16th roots of unity
  u32 x[16];
  // round 1: layers l0 .. l0 + 3; my values: middle index (j << 4) | q
#pragma unroll
  for (int j = 0; j < 16; j++) x[j] = col[base + ((size_t)((j << 4) | q) << sbits) + lo];
  r16_round(x, tw[(((size_t)q << sbits) | (lo0 + lo)) << l0], C);
#pragma unroll
  for (int j = 0; j < 16; j++) sh[((j << 4) | q) * LO + lo + j * LO] = x[j];  // LO words of padding per 16 middle values
  __syncthreads();
  // round 2: layers l0 + 4 .. l0 + 7; my values: middle index (q << 4) | k
#pragma unroll
  for (int k = 0; k < 16; k++) x[k] = sh[((q << 4) | k) * LO + lo + q * LO];
  r16_round(x, tw[(size_t)(lo0 + lo) << (l0 + 4)], C);
#pragma unroll
  for (int k = 0; k < 16; k++) col[base + ((size_t)((q << 4) | k) << sbits) + lo] = x[k];
}
// last layers [l0, n) on contiguous tiles of `ts` = min(4096, 2^n) elements (whole groups of 2^(n - l0) elements)
__global__ __launch_bounds__(256) void ntt_contig_k(u32* __restrict__ data, size_t ld, unsigned n, unsigned l0, unsigned log_ts, const u32* __restrict__ tw) {
  __shared__ u32 s[4096];
  u32* col = data + (size_t)blockIdx.y * ld;
  const unsigned ts = 1u << log_ts;
  const size_t base = (size_t)blockIdx.x << log_ts;
  for (unsigned e = threadIdx.x; e < ts; e += 256) s[e] = col[base + e];
  __syncthreads();
  for (unsigned l = l0; l < n; l++) {
    const unsigned lh = n - l - 1;  // log2 of the butterfly distance
    for (unsigned b = threadIdx.x; b < ts / 2; b += 256) {
      unsigned lo = b & ((1u << lh) - 1), hi = b >> lh;
      unsigned e0 = (hi << (lh + 1)) | lo, e1 = e0 | (1u << lh);
      u32 x = s[e0], y = s[e1];
      u32 w = tw[(size_t)lo << l];
      s[e0] = bb_add(x, y);
      s[e1] = bb_mul(bb_sub(x, y), w);
    }
    __syncthreads();
  }
  for (unsigned e = threadIdx.x; e < ts; e += 256) col[base + e] = s[e];
}
void bb_dif(Ctx& ctx, u32* data, size_t ld, unsigned n, size_t ncols) {
  if (n == 0 || ncols == 0) return;
  if (n > BB_TWO_ADICITY) throw std::runtime_error("transform larger than the two-adicity of BabyBear");
  const u32* tw = twiddles(ctx, n);
  unsigned l0 = 0;
  const bool lds_passes = getenv("MSBB_NTT_LDS") != nullptr;  // the plain LDS radix-2 passes (tests)
  while (!lds_passes && n - l0 >= 12) {  // eight layers per register pass (the low index keeps >= 4 bits: 64-byte runs)
    ProfScope prof(ctx, msamd::K_NTT8S_DIF, 8.0 * double(size_t(1) << n) * double(ncols));
    if (n - l0 >= 13) {
      dim3 grid((unsigned)((size_t(1) << n) >> 13), (unsigned)ncols);
      ntt_r16_k<32><<<grid, 512, 0, ctx.stream>>>(data, ld, n, l0, tw);
    } else {
      dim3 grid((unsigned)((size_t(1) << n) >> 12), (unsigned)ncols);
      ntt_r16_k<16><<<grid, 256, 0, ctx.stream>>>(data, ld, n, l0, tw);
    }
    l0 += 8;
  }
  while (n - l0 > 12) {  // 7 layers per LDS pass
    dim3 grid((unsigned)((size_t(1) << n) >> 12), (unsigned)ncols);
    ProfScope prof(ctx, msamd::K_NTT_STRIDED, 8.0 * double(size_t(1) << n) * double(ncols));
    ntt_strided_k<7, 32><<<grid, 256, 0, ctx.stream>>>(data, ld, n, l0, tw);
    l0 += 7;
  }
  if (l0 == n) return;
  unsigned log_ts = std::min(n, 12u);
  dim3 grid((unsigned)((size_t(1) << n) >> log_ts), (unsigned)ncols);
  ProfScope prof(ctx, msamd::K_NTT_CONTIG, 8.0 * double(size_t(1) << n) * double(ncols));
  ntt_contig_k<<<grid, 256, 0, ctx.stream>>>(data, ld, n, l0, log_ts, tw);
}

// Inverse transform + coset scaling + zero padding in one gather: with X = DFT(x) held bit-reversed (DIF output),
// coefficient k = X[(n - k) mod n] / n; out[k] = coefficient k * shift^k for k < n, 0 for n <= k < N.
__global__ void idft_gather_k(const u32* __restrict__ src, size_t src_ld, unsigned log_n, u32* __restrict__ dst, size_t dst_ld, size_t N,
                              u32 n_inv, u32 shift) {
  size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= N) return;
  const u32* s = src + (size_t)blockIdx.y * src_ld;
  u32* d = dst + (size_t)blockIdx.y * dst_ld;
  size_t n = size_t(1) << log_n;
  if (k >= n) {
    d[k] = 0;
    return;
  }
  size_t j = (n - k) & (n - 1);
  d[k] = bb_mul(bb_mul(s[bitrev_dev(j, log_n)], n_inv), bb_pow(shift, k));
}
void bb_coset_lde(Ctx& ctx, const BMat& evals, unsigned lb, BMat& out) {
  size_t n = evals.h, N = n << lb, w = evals.w;
  unsigned log_n = log2_host(n);
  out = bmat(ctx, N, w);
  if (!w) return;
  DBuf<u32> tmp(ctx, n * w);
  HIP_CHECK(hipMemcpy2DAsync(tmp.p, n * 4, evals.buf.p, evals.ld * 4, n * 4, w, hipMemcpyDeviceToDevice, ctx.stream));
  bb_dif(ctx, tmp.p, n, log_n, w);
  u32 n_inv = bb_inv(bb_to_monty((u32)(n % BB_P)));
  dim3 grid(blocks_for(N, 256), (unsigned)w);
  {
    ProfScope prof(ctx, msamd::K_TRANSPOSE, 4.0 * double(n + N) * double(w));
    idft_gather_k<<<grid, 256, 0, ctx.stream>>>(tmp.p, n, log_n, out.buf.p, out.ld, N, n_inv, bb_to_monty(BB_GENERATOR));
  }
  bb_dif(ctx, out.buf.p, out.ld, log_n + lb, w);  // (tmp returns to the pool: reuse is ordered by the stream)
}

// shifted_quotient_slices + zero padding (src/prover.rs:631-679, 709-717): X = DFT(q) bit-reversed; padded column
// (chunk * 4 + c), row r < n = X_c[(N - (chunk n + r)) mod N] * GENERATOR^(-n chunk) / N
__global__ void quotient_slices_k(const u32* __restrict__ src, size_t src_ld, unsigned log_n, unsigned log_q, u32* __restrict__ dst, size_t dst_ld,
                                  size_t rows_out, u32 weight0, u32 weight_step) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows_out) return;
  unsigned colid = blockIdx.y;  // chunk * 4 + c
  unsigned chunk = colid >> 2, c = colid & 3;
  u32* d = dst + (size_t)colid * dst_ld;
  size_t n = size_t(1) << log_n, N = n << log_q;
  if (r >= n) {
    d[r] = 0;
    return;
  }
  size_t j = (size_t)chunk * n + r;
  size_t srci = bitrev_dev((N - j) & (N - 1), log_n + log_q);
  u32 wgt = bb_mul(weight0, bb_pow(weight_step, chunk));
  d[r] = bb_mul(src[(size_t)c * src_ld + srci], wgt);
}
void bb_quotient_lde(Ctx& ctx, BMat& q_evals, unsigned log_n, unsigned log_q, unsigned lb, BMat& out) {
  size_t n = size_t(1) << log_n, q = size_t(1) << log_q, N = n * q, rows_out = n << lb;
  bb_dif(ctx, q_evals.buf.p, q_evals.ld, log_n + log_q, 4);
  out = bmat(ctx, rows_out, 4 * q);
  u32 g = bb_to_monty(BB_GENERATOR);
  u32 weight0 = bb_inv(bb_to_monty((u32)(N % BB_P)));
  u32 weight_step = bb_inv(bb_pow(g, n));
  dim3 grid(blocks_for(rows_out, 256), (unsigned)(4 * q));
  quotient_slices_k<<<grid, 256, 0, ctx.stream>>>(q_evals.buf.p, q_evals.ld, log_n, log_q, out.buf.p, out.ld, rows_out, weight0, weight_step);
  bb_dif(ctx, out.buf.p, out.ld, log_n + lb, 4 * q);
}

// ------------------------------------------------------------------ Poseidon2 hashing
// PaddingFreeSponge<Perm, 16, 8, 8>: row r of the concatenated columns, 8 words per absorbed block
__global__ __launch_bounds__(256) void leaf_hash_k(const u32* const* __restrict__ cols, unsigned W, size_t rows, const Poseidon2* __restrict__ perm,
                                                   Digest8* __restrict__ out) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  u32 st[16];
#pragma unroll
  for (int i = 0; i < 16; i++) st[i] = 0;
  for (unsigned b = 0; b < W; b += 8) {
#pragma unroll
    for (int i = 0; i < 8; i++)
      if (b + i < W) st[i] = cols[b + i][r];
    bb_poseidon2(*perm, st);
  }
  Digest8 d;
#pragma unroll
  for (int i = 0; i < 8; i++) d.w[i] = st[i];
  out[r] = d;
}
// FRI layer leaves: row i = 8 consecutive words (two E4)
__global__ __launch_bounds__(256) void leaf_hash8_k(const u32* __restrict__ rows8, size_t rows, const Poseidon2* __restrict__ perm, Digest8* __restrict__ out) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  u32 st[16];
  const uint4* p = (const uint4*)(rows8 + r * 8);
  uint4 a = p[0], b = p[1];
  st[0] = a.x, st[1] = a.y, st[2] = a.z, st[3] = a.w, st[4] = b.x, st[5] = b.y, st[6] = b.z, st[7] = b.w;
#pragma unroll
  for (int i = 8; i < 16; i++) st[i] = 0;
  bb_poseidon2(*perm, st);
  Digest8 d;
#pragma unroll
  for (int i = 0; i < 8; i++) d.w[i] = st[i];
  out[r] = d;
}
// FRI layer leaves of a round of arity 2^a >= 4 (max_log_arity > 1): row i = 2^a E4 = nblk blocks of 8 words, absorbed one
// block per permutation (PaddingFreeSponge, rate 8)
__global__ __launch_bounds__(256) void leaf_hash_wide_k(const u32* __restrict__ words, size_t rows, unsigned nblk, const Poseidon2* __restrict__ perm,
                                                        Digest8* __restrict__ out) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  u32 st[16];
#pragma unroll
  for (int i = 0; i < 16; i++) st[i] = 0;
  const uint4* p = (const uint4*)(words + r * 8 * nblk);
  for (unsigned b = 0; b < nblk; b++) {
    uint4 x = p[2 * b], y = p[2 * b + 1];
    st[0] = x.x, st[1] = x.y, st[2] = x.z, st[3] = x.w, st[4] = y.x, st[5] = y.y, st[6] = y.z, st[7] = y.w;
    bb_poseidon2(*perm, st);
  }
  Digest8 d;
#pragma unroll
  for (int i = 0; i < 8; i++) d.w[i] = st[i];
  out[r] = d;
}
// TruncatedPermutation<Perm, 2, 8, 16>: parent = permute(left || right)[..8]; with injection of a shorter matrix's
// row digest: parent = compress(parent, inject[i])
__global__ __launch_bounds__(256) void compress_k(const Digest8* __restrict__ prev, size_t n_out, const Digest8* __restrict__ inject,
                                                  const Poseidon2* __restrict__ perm, Digest8* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  u32 st[16];
  Digest8 l = prev[2 * i], r = prev[2 * i + 1];
#pragma unroll
  for (int k = 0; k < 8; k++) st[k] = l.w[k], st[8 + k] = r.w[k];
  bb_poseidon2(*perm, st);
  if (inject) {
    Digest8 x = inject[i];
#pragma unroll
    for (int k = 0; k < 8; k++) st[8 + k] = x.w[k];
    bb_poseidon2(*perm, st);
  }
  Digest8 d;
#pragma unroll
  for (int k = 0; k < 8; k++) d.w[k] = st[k];
  out[i] = d;
}
__global__ void permute_batch_k(u32* states, size_t n, const Poseidon2* __restrict__ perm) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 st[16];
#pragma unroll
  for (int k = 0; k < 16; k++) st[k] = states[16 * i + k];
  bb_poseidon2(*perm, st);
#pragma unroll
  for (int k = 0; k < 16; k++) states[16 * i + k] = st[k];
}
void bb_permute_batch(Ctx& ctx, const Poseidon2* d_perm, u32* d_states, size_t n) {
  if (n) permute_batch_k<<<blocks_for(n, 256), 256, 0, ctx.stream>>>(d_states, n, d_perm);
}


// ---- one permutation spread over 16 lanes (lane l holds state word l; a DPP row is exactly 16 lanes). A permutation
// done by one lane is a dependent chain of ~10k instructions (~25 us): fine when a layer has > 10^5 nodes to overlap,
// far too slow for the small upper layers of every tree, which are pure latency. Spread out, the chain is ~1.2k
// instructions: the S-boxes of a full round run in parallel, M4 and the column sums are quad / row rotations.
template <int CTRL>
__device__ __forceinline__ u32 dppx(u32 v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
#else
  return v;
#endif
}
__device__ __forceinline__ u32 coop_mds(u32 s) {
  u32 nxt = dppx<0x39>(s);               // quad_perm [1,2,3,0]: the next word of my 4-chunk
  u32 t = bb_add(s, dppx<0xB1>(s));      // quad_perm [1,0,3,2]
  u32 sum = bb_add(t, dppx<0x4E>(t));    // quad_perm [2,3,0,1]: chunk sum
  u32 v = bb_add(bb_add(sum, s), bb_add(nxt, nxt));  // row r of M4: sum + s_r + 2 s_(r+1)
  u32 c = bb_add(v, dppx<0x128>(v));     // row_ror:8
  c = bb_add(c, dppx<0x124>(c));         // row_ror:4 -> sum of the four chunks' word r
  return bb_add(v, c);
}
__device__ __forceinline__ u32 coop_poseidon2(const Poseidon2& k, u32 s, int l) {
  s = coop_mds(s);
#pragma unroll
  for (int r = 0; r < 4; r++) s = coop_mds(bb_sbox7(bb_add(s, k.external[r][l])));
  const u32 dg = k.diag[l];
#pragma unroll
  for (int r = 0; r < 13; r++) {
    u32 x = bb_sbox7(bb_add(s, k.internal[r]));
    s = l == 0 ? x : s;
    u32 t = bb_add(s, dppx<0x128>(s));
    t = bb_add(t, dppx<0x124>(t));
    t = bb_add(t, dppx<0x122>(t));
    t = bb_add(t, dppx<0x121>(t));  // every lane: the sum of the 16 words
    s = bb_add(t, bb_mul(dg, s));
  }
#pragma unroll
  for (int r = 4; r < 8; r++) s = coop_mds(bb_sbox7(bb_add(s, k.external[r][l])));
  return s;
}
__global__ __launch_bounds__(256) void compress_coop_k(const Digest8* __restrict__ prev, size_t n_out, const Digest8* __restrict__ inject,
                                                       const Poseidon2* __restrict__ perm, Digest8* __restrict__ out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t node = t >> 4;
  int l = (int)(t & 15);
  if (node >= n_out) return;
  u32 s = ((const u32*)prev)[node * 16 + l];  // left || right are adjacent digests
  s = coop_poseidon2(*perm, s, l);
  if (inject) {
    if (l >= 8) s = inject[node].w[l - 8];
    s = coop_poseidon2(*perm, s, l);
  }
  if (l < 8) out[node].w[l] = s;
}
__global__ __launch_bounds__(256) void leaf_hash_coop_k(const u32* const* __restrict__ cols, unsigned W, size_t rows, const Poseidon2* __restrict__ perm,
                                                        Digest8* __restrict__ out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t r = t >> 4;
  int l = (int)(t & 15);
  if (r >= rows) return;
  u32 s = 0;
  for (unsigned b = 0; b < W; b += 8) {
    if (l < 8 && b + l < W) s = cols[b + l][r];
    s = coop_poseidon2(*perm, s, l);
  }
  if (l < 8) out[r].w[l] = s;
}
__global__ __launch_bounds__(256) void leaf_hash8_coop_k(const u32* __restrict__ rows8, size_t rows, const Poseidon2* __restrict__ perm, Digest8* __restrict__ out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t r = t >> 4;
  int l = (int)(t & 15);
  if (r >= rows) return;
  u32 s = l < 8 ? rows8[r * 8 + l] : 0;
  s = coop_poseidon2(*perm, s, l);
  if (l < 8) out[r].w[l] = s;
}
// the top of a tree in one launch: levels[0] (n0 <= 128 digests) -> levels[1] -> ... -> a single digest
struct TailArgs {
  Digest8* level[9];
  int n_levels;  // level[0] is the input; level[k] has n0 >> k digests
  unsigned n0;
};
__global__ __launch_bounds__(1024) void tree_tail_k(TailArgs a, const Poseidon2* __restrict__ perm) {
  const int l = threadIdx.x & 15;
  const unsigned group = threadIdx.x >> 4;  // 64 groups
  for (int k = 1; k < a.n_levels; k++) {
    unsigned n_out = a.n0 >> k;
    if (group < n_out) {  // n_out <= 64
      u32 s = ((const u32*)a.level[k - 1])[group * 16 + l];
      s = coop_poseidon2(*perm, s, l);
      if (l < 8) a.level[k][group].w[l] = s;
    }
    __threadfence_block();
    __syncthreads();
  }
}
static constexpr size_t COOP_MAX = size_t(1) << 15;  // below this many nodes a layer cannot hide the single-lane latency

static void hash_group(Ctx& ctx, const Poseidon2* d_perm, const std::vector<const BMat*>& group, size_t rows, Digest8* out) {
  std::vector<const u32*> cols;
  for (auto m : group)
    for (size_t c = 0; c < m->w; c++) cols.push_back(m->col(c));
  DBuf<const u32*> d_cols(ctx, std::max<size_t>(cols.size(), 1));
  if (!cols.empty()) ctx.h2d(d_cols.p, cols.data(), cols.size() * sizeof(const u32*));
  // PaddingFreeSponge<_, 16, 8, 8>: one permutation per 8 row elements
  ProfScope prof(ctx, msamd::K_LEAF_HASH, double(rows) * (4.0 * cols.size() + 32.0), double(rows) * double(std::max<size_t>((cols.size() + 7) / 8, 1)));
  if (rows <= COOP_MAX)
    leaf_hash_coop_k<<<blocks_for(rows * 16, 256), 256, 0, ctx.stream>>>(d_cols.p, (unsigned)cols.size(), rows, d_perm, out);
  else
    leaf_hash_k<<<blocks_for(rows, 256), 256, 0, ctx.stream>>>(d_cols.p, (unsigned)cols.size(), rows, d_perm, out);
}
// every remaining level of t (whose last layer has `cur` <= 2^16 digests, no injection left) in one launch; with `chp` the
// launch ends with the round's challenger step. Defined next to the kernel (subtree_k) further down.
struct FriRoundCh {
  DevChallenger* ch;
  FriBeta* beta_out;
};
static void launch_subtree(Ctx& ctx, const Poseidon2* d_perm, BTree& t, const FriRoundCh* chp);
// A sixteen-lane Poseidon2 permutation is ~1.2 k dependent instructions (2 us on a SIMD of its own, 8 us when four waves
// share it), so - unlike the BLAKE3 trees of the Goldilocks path - a level with more than 64 nodes is faster as its own
// chip-wide launch than inside one workgroup (measured: a 2^16-child tree 86 us in one launch against ~70 us layer by layer).
// The one-launch form therefore takes over only where a level fits one pass of the 64 groups; MSBB_SUBTREE_MAX_LOG moves it.
static size_t subtree_max() {
  const char* e = getenv("MSBB_SUBTREE_MAX_LOG");
  return size_t(1) << (e ? atoi(e) : 7);
}

static void build_upper_layers(Ctx& ctx, const Poseidon2* d_perm, BTree& t, const std::vector<const BMat*>& order, size_t pos,
                               const FriRoundCh* chp = nullptr) {
  const bool fused_ok = !getenv("MSBB_NO_SUBTREE");
  while (t.sizes.back() > 1) {
    size_t cur = t.sizes.back(), nl = cur / 2;
    if (fused_ok && pos == order.size() && cur <= subtree_max()) {
      const bool with_ch = chp && t.cap_height == 0;  // a cap of several digests is observed by the separate step below
      launch_subtree(ctx, d_perm, t, with_ch ? chp : nullptr);
      if (with_ch) chp = nullptr;
      break;
    }
    if (pos == order.size() && cur <= 128) {  // nothing left to inject: the rest of the tree in one launch
      TailArgs a;
      a.n0 = (unsigned)cur;
      a.level[0] = t.layers.back().p;
      a.n_levels = 1;
      for (size_t n = nl; n >= 1; n /= 2) {
        t.layers.emplace_back(ctx, n);
        t.sizes.push_back(n);
        a.level[a.n_levels++] = t.layers.back().p;
        if (n == 1) break;
      }
      ProfScope prof(ctx, msamd::K_COMPRESS, 96.0 * double(cur), double(cur));
      tree_tail_k<<<1, 1024, 0, ctx.stream>>>(a, d_perm);
      break;
    }
    std::vector<const BMat*> group;
    while (pos < order.size() && order[pos]->h == nl) group.push_back(order[pos++]);
    DBuf<Digest8> inj;
    if (!group.empty()) {
      inj = DBuf<Digest8>(ctx, nl);
      hash_group(ctx, d_perm, group, nl, inj.p);
    }
    DBuf<Digest8> next(ctx, nl);
    ProfScope prof(ctx, msamd::K_COMPRESS, 96.0 * double(nl), double(nl) * (group.empty() ? 1.0 : 2.0));
    if (nl <= COOP_MAX)
      compress_coop_k<<<blocks_for(nl * 16, 256), 256, 0, ctx.stream>>>(t.layers.back().p, nl, group.empty() ? nullptr : inj.p, d_perm, next.p);
    else
      compress_k<<<blocks_for(nl, 256), 256, 0, ctx.stream>>>(t.layers.back().p, nl, group.empty() ? nullptr : inj.p, d_perm, next.p);
    t.layers.push_back(std::move(next));
    t.sizes.push_back(nl);
  }
  if (pos != order.size()) throw std::runtime_error("mmcs commit: a matrix height was never reached");
  if (chp) {  // the tree was finished by the layer-by-layer kernels (or is a single leaf): the challenger step on its own
    const size_t cl = t.cap_layer();
    bb_fri_challenge(ctx, chp->ch, t.layers[cl].p, t.sizes[cl], d_perm, chp->beta_out);
  }
}
void bb_commit(Ctx& ctx, const Poseidon2* d_perm, std::vector<BMat>&& ldes, unsigned cap_height, BPcsData& out) {
  out.ldes = std::move(ldes);
  BTree& t = out.tree;
  t = BTree();
  t.cap_height = cap_height;
  if (out.ldes.empty()) throw std::runtime_error("mmcs commit: no matrices");
  std::vector<const BMat*> order;
  for (auto& m : out.ldes) {
    if (m.h == 0 || (m.h & (m.h - 1))) throw std::runtime_error("mmcs commit: heights must be powers of two");
    order.push_back(&m);
  }
  std::stable_sort(order.begin(), order.end(), [](const BMat* a, const BMat* b) { return a->h > b->h; });
  size_t pos = 0, maxh = order[0]->h;
  std::vector<const BMat*> group;
  while (pos < order.size() && order[pos]->h == maxh) group.push_back(order[pos++]);
  t.layers.emplace_back(ctx, maxh);
  t.sizes.push_back(maxh);
  hash_group(ctx, d_perm, group, maxh, t.layers[0].p);
  build_upper_layers(ctx, d_perm, t, order, pos);
}
void bb_commit_pairs(Ctx& ctx, const Poseidon2* d_perm, const E4* d_vec, size_t rows, unsigned cap_height, BTree& t, DevChallenger* d_ch,
                     FriBeta* d_beta_out, unsigned log_arity) {
  if (log_arity < 1 || log_arity > BB_FRI_MAX_LOG_ARITY) throw std::runtime_error("FRI: round arity out of range");
  t = BTree();
  t.cap_height = cap_height;
  t.layers.emplace_back(ctx, rows);
  t.sizes.push_back(rows);
  {
    ProfScope prof(ctx, msamd::K_LEAF_HASH, (32.0 + 16.0 * double(1u << log_arity)) * double(rows), double(rows) * double(1u << (log_arity - 1)));
    if (log_arity > 1)
      leaf_hash_wide_k<<<blocks_for(rows, 256), 256, 0, ctx.stream>>>((const u32*)d_vec, rows, 1u << (log_arity - 1), d_perm, t.layers[0].p);
    else if (rows <= COOP_MAX)
      leaf_hash8_coop_k<<<blocks_for(rows * 16, 256), 256, 0, ctx.stream>>>((const u32*)d_vec, rows, d_perm, t.layers[0].p);
    else
      leaf_hash8_k<<<blocks_for(rows, 256), 256, 0, ctx.stream>>>((const u32*)d_vec, rows, d_perm, t.layers[0].p);
  }
  FriRoundCh chp{d_ch, d_beta_out};
  build_upper_layers(ctx, d_perm, t, {}, 0, d_ch ? &chp : nullptr);
}

// ------------------------------------------------------------------ node programs
void bb_build_program(Ctx& ctx, const std::vector<PNode>& nodes, BProgram& out) {
  size_t n = nodes.size();
  std::vector<u32> k(n), a(n), b(n);
  for (size_t i = 0; i < n; i++) {
    k[i] = nodes[i].kind | (nodes[i].source << 8) | (nodes[i].offset << 16);
    a[i] = nodes[i].kind == msamd::OP_CONST ? bb_to_monty((u32)nodes[i].a) : (u32)nodes[i].a;
    b[i] = (u32)nodes[i].b;
  }
  out.n = n;
  out.kind = DBuf<u32>(ctx, std::max<size_t>(n, 1));
  out.a = DBuf<u32>(ctx, std::max<size_t>(n, 1));
  out.b = DBuf<u32>(ctx, std::max<size_t>(n, 1));
  if (n) {
    ctx.h2d(out.kind.p, k.data(), n * 4);
    ctx.h2d(out.a.p, a.data(), n * 4);
    ctx.h2d(out.b.p, b.data(), n * 4);
    ctx.sync();
  }
}

// One sweep of the node program (src/eval.rs:36-111) for one row; node values go to a per-thread slot column in global
// scratch (slot s of thread t at scratch[s * stride + t]: coalesced across the wave).
struct RowView {
  const u32 *pre0, *pre1, *main0, *main1, *s20, *s21;  // element c of a row at ptr[c * ld]
  size_t pre_ld, main_ld, s2_ld;
  const u32* publics;  // 16 base coordinates (may be null)
  u32 is_first, is_last, is_trans;
};
__device__ __forceinline__ void sweep(const u32* __restrict__ kind, const u32* __restrict__ na, const u32* __restrict__ nb, unsigned len, const RowView& v,
                                      u32* __restrict__ slots, size_t stride) {
  for (unsigned i = 0; i < len; i++) {
    u32 k = kind[i], a = na[i], val;
    switch (k & 0xff) {
      case msamd::OP_CONST: val = a; break;
      case msamd::OP_VAR: {
        unsigned src = (k >> 8) & 0xff, off = (k >> 16) & 0xff;
        if (src == 0)
          val = (off ? v.pre1 : v.pre0)[(size_t)a * v.pre_ld];
        else if (src == 1)
          val = (off ? v.main1 : v.main0)[(size_t)a * v.main_ld];
        else
          val = (off ? v.s21 : v.s20)[(size_t)a * v.s2_ld];
        break;
      }
      case msamd::OP_PUBLIC: val = v.publics ? v.publics[a] : 0; break;
      case msamd::OP_IS_FIRST: val = v.is_first; break;
      case msamd::OP_IS_LAST: val = v.is_last; break;
      case msamd::OP_IS_TRANS: val = v.is_trans; break;
      case msamd::OP_ADD: val = bb_add(slots[(size_t)a * stride], slots[(size_t)nb[i] * stride]); break;
      case msamd::OP_SUB: val = bb_sub(slots[(size_t)a * stride], slots[(size_t)nb[i] * stride]); break;
      case msamd::OP_MUL: val = bb_mul(slots[(size_t)a * stride], slots[(size_t)nb[i] * stride]); break;
      default: val = bb_neg(slots[(size_t)a * stride]); break;
    }
    slots[(size_t)i * stride] = val;
  }
}

// ------------------------------------------------------------------ stage 2 (src/lookup.rs:472-555)
struct Stage2Args {
  const u32 *kind, *na, *nb;
  unsigned prefix_len;
  const u32 *lk_mult, *lk_off, *lk_args;
  unsigned L;
  const u32* trace;
  size_t trace_ld;
  const u32* pre;
  size_t pre_ld;
  size_t n;
  E4 beta, gamma;
  u32* scratch;
  E4* terms;  // n * L, row-major (row, lookup)
};
__global__ __launch_bounds__(256) void stage2_terms_k(Stage2Args p) {
  size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= p.n) return;
  size_t rn = r + 1 == p.n ? 0 : r + 1;
  RowView v;
  v.pre0 = p.pre ? p.pre + r : nullptr, v.pre1 = p.pre ? p.pre + rn : nullptr, v.pre_ld = p.pre_ld;
  v.main0 = p.trace + r, v.main1 = p.trace + rn, v.main_ld = p.trace_ld;
  v.s20 = v.s21 = nullptr, v.s2_ld = 0;
  v.publics = nullptr;
  v.is_first = r == 0 ? BB_R1 : 0;
  v.is_last = r + 1 == p.n ? BB_R1 : 0;
  v.is_trans = r + 1 == p.n ? 0 : BB_R1;
  u32* slots = p.scratch + r;
  sweep(p.kind, p.na, p.nb, p.prefix_len, v, slots, p.n);
  for (unsigned l = 0; l < p.L; l++) {
    E4 f = e4_zero();
    for (unsigned k = p.lk_off[l + 1]; k-- > p.lk_off[l];) {  // Horner over the reversed arguments, src/lookup.rs:375-384
      f = e4_mul(f, p.gamma);
      f.c[0] = bb_add(f.c[0], slots[(size_t)p.lk_args[k] * p.n]);
    }
    E4 msg = e4_add(f, p.beta);
    p.terms[r * p.L + l] = e4_mul_base(e4_inv(msg), slots[(size_t)p.lk_mult[l] * p.n]);
  }
}
// exclusive prefix sums of E4 in three steps: per-block (1024 elements) local scan + block totals, serial scan of the
// totals (one thread: a few thousand additions), write-out with the block offsets into the stage-2 columns
__global__ __launch_bounds__(256) void scan_local_k(E4* __restrict__ v, size_t n, E4* __restrict__ block_tot) {
  __shared__ E4 s[256];
  size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x * 4;
  E4 x[4], run = e4_zero();
#pragma unroll
  for (int i = 0; i < 4; i++) {
    x[i] = base + i < n ? v[base + i] : e4_zero();
    E4 t = x[i];
    x[i] = run;
    run = e4_add(run, t);
  }
  s[threadIdx.x] = run;
  __syncthreads();
  for (unsigned d = 1; d < 256; d <<= 1) {  // Hillis-Steele inclusive scan of the thread totals
    E4 t = threadIdx.x >= d ? s[threadIdx.x - d] : e4_zero();
    __syncthreads();
    s[threadIdx.x] = e4_add(s[threadIdx.x], t);
    __syncthreads();
  }
  E4 off = threadIdx.x ? s[threadIdx.x - 1] : e4_zero();
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (base + i < n) v[base + i] = e4_add(x[i], off);
  if (threadIdx.x == 255) block_tot[blockIdx.x] = s[255];
}
__global__ __launch_bounds__(1024) void scan_totals_k(E4* block_tot, size_t nb, E4* total) {
  __shared__ E4 s[1024];
  __shared__ E4 carry;
  if (threadIdx.x == 0) carry = e4_zero();
  __syncthreads();
  for (size_t base = 0; base < nb; base += 1024) {
    size_t i = base + threadIdx.x;
    E4 mine = i < nb ? block_tot[i] : e4_zero();
    s[threadIdx.x] = mine;
    __syncthreads();
    for (unsigned d = 1; d < 1024; d <<= 1) {
      E4 t = threadIdx.x >= d ? s[threadIdx.x - d] : e4_zero();
      __syncthreads();
      s[threadIdx.x] = e4_add(s[threadIdx.x], t);
      __syncthreads();
    }
    E4 c = carry;
    if (i < nb) block_tot[i] = e4_add(c, e4_sub(s[threadIdx.x], mine));  // exclusive
    __syncthreads();
    if (threadIdx.x == 1023) carry = e4_add(c, s[1023]);
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}
__global__ void stage2_write_k(const E4* __restrict__ v, const E4* __restrict__ block_off, size_t n_rows, unsigned L, u32* __restrict__ out, size_t ld) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * L) return;
  size_t r = idx / L, l = idx % L;
  E4 x = e4_add(v[idx], block_off[idx >> 10]);
#pragma unroll
  for (int k = 0; k < 4; k++) out[(4 * l + k) * ld + r] = x.c[k];
}
void bb_stage2(Ctx& ctx, const BProgram& prog, size_t prefix_len, const BLookupsDev& lk, const BMat& trace, const BMat* pre, E4 beta, E4 gamma,
               BMat& out, E4* total) {
  size_t n = trace.h, L = lk.L;
  out = bmat(ctx, n, 4 * std::max<size_t>(L, 1));
  *total = e4_zero();
  if (L == 0) {  // pass-through accumulator column: zeros (src/lookup.rs:520-523)
    HIP_CHECK(hipMemsetAsync(out.buf.p, 0, n * 4 * 4, ctx.stream));
    return;
  }
  DBuf<u32> scratch(ctx, std::max<size_t>(prefix_len, 1) * n);
  DBuf<E4> terms(ctx, n * L);
  size_t nb = (n * L + 1023) / 1024;
  DBuf<E4> tot(ctx, nb + 1);
  Stage2Args a;
  a.kind = prog.kind.p, a.na = prog.a.p, a.nb = prog.b.p, a.prefix_len = (unsigned)prefix_len;
  a.lk_mult = lk.mult.p, a.lk_off = lk.arg_off.p, a.lk_args = lk.args.p, a.L = (unsigned)L;
  a.trace = trace.buf.p, a.trace_ld = trace.ld;
  a.pre = pre ? pre->buf.p : nullptr, a.pre_ld = pre ? pre->ld : 0;
  a.n = n, a.beta = beta, a.gamma = gamma, a.scratch = scratch.p, a.terms = terms.p;
  stage2_terms_k<<<blocks_for(n, 256), 256, 0, ctx.stream>>>(a);
  scan_local_k<<<(unsigned)nb, 256, 0, ctx.stream>>>(terms.p, n * L, tot.p);
  scan_totals_k<<<1, 1024, 0, ctx.stream>>>(tot.p, nb, tot.p + nb);
  stage2_write_k<<<blocks_for(n * L, 256), 256, 0, ctx.stream>>>(terms.p, tot.p, n, (unsigned)L, out.buf.p, out.ld);
  ctx.d2h(total, tot.p + nb, sizeof(E4));
}

// claims accumulator (src/prover.rs:382-387): sum over the claims of 1 / (beta + fingerprint(gamma, claim))
__global__ void claims_terms_k(const u32* __restrict__ data, const u64* __restrict__ offs, size_t n, E4 beta, E4 gamma, E4* __restrict__ terms) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  E4 f = e4_zero();
  for (u64 k = offs[i + 1]; k-- > offs[i];) {
    f = e4_mul(f, gamma);
    f.c[0] = bb_add(f.c[0], data[k]);
  }
  terms[i] = e4_inv(e4_add(beta, f));
}
E4 bb_claims_accumulator(Ctx& ctx, const u32* d_data_monty, const u64* d_offs, size_t n, E4 beta, E4 gamma) {
  if (!n) return e4_zero();
  DBuf<E4> terms(ctx, n);
  size_t nb = (n + 1023) / 1024;
  DBuf<E4> tot(ctx, nb + 1);
  claims_terms_k<<<blocks_for(n, 256), 256, 0, ctx.stream>>>(d_data_monty, d_offs, n, beta, gamma, terms.p);
  scan_local_k<<<(unsigned)nb, 256, 0, ctx.stream>>>(terms.p, n, tot.p);
  scan_totals_k<<<1, 1024, 0, ctx.stream>>>(tot.p, nb, tot.p + nb);
  E4 total;
  ctx.d2h(&total, tot.p + nb, sizeof(E4));
  return total;
}

// DuplexChallenger::grind: the smallest w with sample_bits(bits) == 0 after observe(w). With `pending` values queued,
// the trial is: overwrite state[..pending] with them, state[pending] = w, permute, take the last rate word.
struct GrindArgs {
  u32 state[16];  // sponge state with the queued values already written over its first words
  u32 slot;       // where the witness goes
  u32 mask;
  u32 base, count;
};
__global__ __launch_bounds__(256) void grind_k(GrindArgs a, const Poseidon2* __restrict__ perm, u32* __restrict__ best) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.count) return;
  u32 w = a.base + i;
  u32 st[16];
#pragma unroll
  for (int k = 0; k < 16; k++) st[k] = a.state[k];
#pragma unroll
  for (int k = 0; k < 8; k++)
    if ((u32)k == a.slot) st[k] = bb_to_monty(w);
  bb_poseidon2(*perm, st);
  if ((bb_from_monty(st[7]) & a.mask) == 0) atomicMin(best, w);
}
u32 bb_grind(Ctx& ctx, const Poseidon2* d_perm, const u32* state16, const u32* pending, unsigned n_pending, unsigned bits) {
  GrindArgs a;
  for (int k = 0; k < 16; k++) a.state[k] = state16[k];
  for (unsigned k = 0; k < n_pending; k++) a.state[k] = pending[k];
  a.slot = n_pending;
  a.mask = (1u << bits) - 1;
  DBuf<u32> best(ctx, 1);
  const u32 batch = 1u << 18;
  for (u64 base = 0; base < BB_P; base += batch) {
    u32 init = 0xffffffffu, found;
    ctx.h2d(best.p, &init, 4);
    a.base = (u32)base;
    a.count = (u32)std::min<u64>(batch, BB_P - base);
    grind_k<<<blocks_for(a.count, 256), 256, 0, ctx.stream>>>(a, d_perm, best.p);
    ctx.d2h(&found, best.p, 4);
    if (found != 0xffffffffu) return found;
  }
  throw std::runtime_error("grind: no witness");
}

// ------------------------------------------------------------------ quotient (src/prover.rs:756-962)
__global__ __launch_bounds__(256) void quotient_k(QuotArgs p) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.rows) return;
  const unsigned log_big = p.log_n + p.log_q;
  const size_t N = size_t(1) << log_big, q = size_t(1) << p.log_q;
  size_t st = p.row0 + t;               // storage row of the LDEs
  size_t i = bitrev_dev(st, log_big);   // natural index on the quotient domain GENERATOR * H_N
  size_t st_next = bitrev_dev((i + q) & (N - 1), log_big);
  // selectors at x = g w^i (p3 selectors_on_coset; normalisation pinned by src/lookup.rs:697-756)
  u32 x = bb_mul(p.g, bb_pow(p.w_big, i));
  u32 zh = bb_sub(bb_mul(p.g_pow_n, bb_pow(p.w_q, i & (q - 1))), BB_R1);
  u32 d1 = bb_sub(x, BB_R1), d2 = bb_sub(x, p.gn_inv);
  u32 d12 = bb_mul(d1, d2);
  u32 all_inv = bb_inv(bb_mul(d12, zh));
  u32 inv_zh = bb_mul(all_inv, d12);
  u32 inv12 = bb_mul(all_inv, zh);
  u32 inv_d1 = bb_mul(inv12, d2), inv_d2 = bb_mul(inv12, d1);
  RowView v;
  v.pre0 = p.pre ? p.pre + st : nullptr, v.pre1 = p.pre ? p.pre + st_next : nullptr, v.pre_ld = p.pre_ld;
  v.main0 = p.s1 + st, v.main1 = p.s1 + st_next, v.main_ld = p.s1_ld;
  v.s20 = p.s2 + st, v.s21 = p.s2 + st_next, v.s2_ld = p.s2_ld;
  v.publics = p.publics;
  v.is_first = bb_mul(zh, inv_d1);
  v.is_last = bb_mul(zh, inv_d2);
  v.is_trans = d2;
  u32* slots = p.scratch + t;
  sweep(p.kind, p.na, p.nb, p.n_nodes, v, slots, p.stride);
  E4 acc = e4_zero();
  unsigned cj = 0;
  for (unsigned z = 0; z < p.n_zeros; z++) acc = e4_add(acc, e4_mul_base(p.apow[cj++], slots[(size_t)p.zeros[z] * p.stride]));
  // logUp constraints (src/lookup.rs:152-256): on this domain every working value is a base element, so the coordinate
  // products of the reference are the products of E4 values assembled from those coordinates
  E4 beta = E4{{p.publics[0], p.publics[1], p.publics[2], p.publics[3]}};
  E4 gamma = E4{{p.publics[4], p.publics[5], p.publics[6], p.publics[7]}};
  E4 inj;
#pragma unroll
  for (int k = 0; k < 4; k++) inj.c[k] = bb_mul(v.is_last, p.delta[k]);
  auto s2_at = [&](const u32* rowp, unsigned slot) {
    E4 e;
#pragma unroll
    for (int k = 0; k < 4; k++) e.c[k] = rowp[(size_t)(4 * slot + k) * p.s2_ld];
    return e;
  };
  if (p.L == 0) {
    E4 c = e4_add(e4_sub(s2_at(v.s21, 0), s2_at(v.s20, 0)), inj);
#pragma unroll
    for (int k = 0; k < 4; k++) acc = e4_add(acc, e4_mul_base(p.apow[cj++], c.c[k]));
  } else {
    for (unsigned j = 0; j < p.L; j++) {
      E4 src = s2_at(v.s20, j);
      E4 tgt = j + 1 < p.L ? s2_at(v.s20, j + 1) : e4_add(s2_at(v.s21, 0), inj);
      E4 f = e4_zero();
      for (unsigned k = p.lk_off[j + 1]; k-- > p.lk_off[j];) {
        f = e4_mul(f, gamma);
        f.c[0] = bb_add(f.c[0], slots[(size_t)p.lk_args[k] * p.stride]);
      }
      E4 c = e4_mul(e4_add(f, beta), e4_sub(tgt, src));
      c.c[0] = bb_sub(c.c[0], slots[(size_t)p.lk_mult[j] * p.stride]);
#pragma unroll
      for (int k = 0; k < 4; k++) acc = e4_add(acc, e4_mul_base(p.apow[cj++], c.c[k]));
    }
  }
  acc = e4_mul_base(acc, inv_zh);
#pragma unroll
  for (int k = 0; k < 4; k++) p.out[(size_t)k * p.out_ld + i] = acc.c[k];
}
void bb_quotient(Ctx& ctx, const BQuotientIn& in, BMat& q_evals) {
  unsigned log_big = in.log_n + in.log_q;
  size_t N = size_t(1) << log_big, n = size_t(1) << in.log_n;
  q_evals = bmat(ctx, N, 4);
  // reversed alpha powers
  std::vector<E4> apow(in.constraint_count);
  E4 a = e4_one();
  for (size_t i = 0; i < in.constraint_count; i++) {
    apow[in.constraint_count - 1 - i] = a;
    a = e4_mul(a, in.alpha);
  }
  DBuf<E4> d_apow(ctx, std::max<size_t>(apow.size(), 1));
  if (!apow.empty()) ctx.h2d(d_apow.p, apow.data(), apow.size() * sizeof(E4));
  QuotArgs p;
  p.kind = in.prog->kind.p, p.na = in.prog->a.p, p.nb = in.prog->b.p, p.n_nodes = (unsigned)in.prog->n;
  p.zeros = in.d_zeros, p.n_zeros = (unsigned)in.n_zeros;
  p.lk_mult = in.lk->mult.p, p.lk_off = in.lk->arg_off.p, p.lk_args = in.lk->args.p, p.L = (unsigned)in.lk->L;
  p.pre = in.pre ? in.pre->buf.p : nullptr, p.pre_ld = in.pre ? in.pre->ld : 0;
  p.s1 = in.s1->buf.p, p.s1_ld = in.s1->ld, p.s2 = in.s2->buf.p, p.s2_ld = in.s2->ld;
  p.log_n = in.log_n, p.log_q = in.log_q;
  for (int k = 0; k < 4; k++)
    for (int d = 0; d < 4; d++) p.publics[4 * k + d] = in.publics[k].c[d];
  u32 g_n = bb_two_adic_generator(in.log_n);
  u32 inj_norm = bb_inv(bb_mul(bb_to_monty((u32)(n % BB_P)), g_n));  // 1 / (n g), src/prover.rs:815-823
  for (int d = 0; d < 4; d++) p.delta[d] = bb_mul(bb_sub(in.publics[3].c[d], in.publics[2].c[d]), inj_norm);
  p.apow = d_apow.p;
  p.out = q_evals.buf.p, p.out_ld = q_evals.ld;
  p.g = bb_to_monty(BB_GENERATOR), p.w_big = bb_two_adic_generator(log_big), p.gn_inv = bb_inv(g_n);
  p.g_pow_n = bb_exp_pow2(p.g, in.log_n), p.w_q = bb_two_adic_generator(in.log_q);
  if (in.jit && in.jit->function && !getenv("MSAMD_NO_JIT")) {
    // the circuit's own kernel (quotient_jit.hip): node values live in registers, no slot file, one launch
    p.scratch = nullptr, p.stride = 0, p.row0 = 0, p.rows = N;
    ProfScope prof(ctx, msamd::K_QUOTIENT, double(N) * (8.0 * double(in.s1->w + in.s2->w + (in.pre ? in.pre->w : 0)) + 16.0));
    msamd::bb_quotient_jit_launch(ctx, *in.jit, &p, sizeof(p), N);
    return;
  }
  // rows in chunks, so that the slot file (nodes x rows words) stays bounded
  size_t chunk = std::min<size_t>(N, std::max<size_t>(256, (size_t(1) << 28) / std::max<size_t>(in.prog->n, 1) / 256 * 256));
  if (const char* e = getenv("MSBB_QUOTIENT_CHUNK")) chunk = std::min<size_t>(N, std::max<size_t>(64, (size_t)atoll(e)));  // tests
  DBuf<u32> scratch(ctx, std::max<size_t>(in.prog->n, 1) * chunk);
  p.scratch = scratch.p, p.stride = chunk;
  for (size_t row0 = 0; row0 < N; row0 += chunk) {
    p.row0 = row0, p.rows = std::min(chunk, N - row0);
    ProfScope prof(ctx, msamd::K_QUOTIENT, double(p.rows) * (8.0 * double(in.s1->w + in.s2->w + (in.pre ? in.pre->w : 0)) + 16.0));
    quotient_k<<<blocks_for(p.rows, 256), 256, 0, ctx.stream>>>(p);
  }
}

// ------------------------------------------------------------------ opening
__global__ void inv_denoms_k(E4 z, unsigned log_h, size_t count, u32 g, u32 w, E4* __restrict__ d_inv, E4* __restrict__ d_wgt) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  u32 x = bb_mul(g, bb_pow(w, bitrev_dev(i, log_h)));
  E4 d = z;
  d.c[0] = bb_sub(d.c[0], x);
  E4 inv = e4_inv(d);
  d_inv[i] = inv;
  if (d_wgt) d_wgt[i] = e4_mul_base(inv, x);
}
void bb_inv_denoms(Ctx& ctx, E4 z, unsigned log_h, size_t count, E4* d_inv, E4* d_wgt) {
  inv_denoms_k<<<blocks_for(count, 256), 256, 0, ctx.stream>>>(z, log_h, count, bb_to_monty(BB_GENERATOR), bb_two_adic_generator(log_h), d_inv, d_wgt);
}
static constexpr size_t BARY_ROWS = 8192;
__global__ __launch_bounds__(256) void bary_partial_k(const u32* __restrict__ m, size_t ld, size_t h, const E4* __restrict__ wgt, E4* __restrict__ part, size_t nchunks) {
  __shared__ E4 s[256];
  const u32* col = m + (size_t)blockIdx.y * ld;
  size_t r0 = (size_t)blockIdx.x * BARY_ROWS, r1 = min(h, r0 + BARY_ROWS);
  E4 acc = e4_zero();
  for (size_t r = r0 + threadIdx.x; r < r1; r += 256) acc = e4_add(acc, e4_mul_base(wgt[r], col[r]));
  s[threadIdx.x] = acc;
  __syncthreads();
  for (unsigned d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d) s[threadIdx.x] = e4_add(s[threadIdx.x], s[threadIdx.x + d]);
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(size_t)blockIdx.y * nchunks + blockIdx.x] = s[0];
}
size_t bb_bary_partials(size_t w, size_t h) { return w * ((h + BARY_ROWS - 1) / BARY_ROWS); }
void bb_bary_launch(Ctx& ctx, const BMat& m, size_t h, const E4* d_wgt, E4* d_part) {
  if (!m.w) return;
  size_t nch = (h + BARY_ROWS - 1) / BARY_ROWS;
  dim3 grid((unsigned)nch, (unsigned)m.w);
  bary_partial_k<<<grid, 256, 0, ctx.stream>>>(m.buf.p, m.ld, h, d_wgt, d_part, nch);
}
void bb_bary_finish(const E4* h_part, size_t w, size_t h, std::vector<E4>& sums) {
  size_t nch = (h + BARY_ROWS - 1) / BARY_ROWS;
  sums.assign(w, e4_zero());
  for (size_t c = 0; c < w; c++)
    for (size_t k = 0; k < nch; k++) sums[c] = e4_add(sums[c], h_part[c * nch + k]);
}
struct DeepArgs {
  const u32* m;
  size_t ld, h;
  unsigned w;
  const E4* apow;
  int npoints;
  const E4* inv[2];
  E4 K[2], off[2];
  E4* ro;
};
__global__ __launch_bounds__(256) void deep_k(DeepArgs p) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.h) return;
  E4 comp = e4_zero();
  for (unsigned c = 0; c < p.w; c++) comp = e4_add(comp, e4_mul_base(p.apow[c], p.m[(size_t)c * p.ld + i]));
  E4 r = p.ro[i];
  for (int k = 0; k < p.npoints; k++) r = e4_add(r, e4_mul(p.inv[k][i], e4_sub(p.K[k], e4_mul(p.off[k], comp))));
  p.ro[i] = r;
}
void bb_deep(Ctx& ctx, const BMat& m, const E4* d_apow, int npoints, const E4* const* d_inv, const E4* K, const E4* off, E4* d_ro) {
  if (npoints == 0 || m.h == 0) return;
  DeepArgs p;
  p.m = m.buf.p, p.ld = m.ld, p.h = m.h, p.w = (unsigned)m.w, p.apow = d_apow, p.npoints = npoints, p.ro = d_ro;
  for (int k = 0; k < 2; k++) {
    p.inv[k] = k < npoints ? d_inv[k] : nullptr;
    p.K[k] = k < npoints ? K[k] : e4_zero();
    p.off[k] = k < npoints ? off[k] : e4_zero();
  }
  ProfScope prof(ctx, msamd::K_DEEP, double(m.h) * (4.0 * double(m.w) + 32.0));
  deep_k<<<blocks_for(m.h, 256), 256, 0, ctx.stream>>>(p);
}
// FRI fold of one layer (p3 TwoAdicFriFolding::fold_matrix, arity 2): rows (lo, hi) are the values at (x, -x) with
// x = w^bitrev(i) of the order-2 rows subgroup; out = (lo + hi) / 2 + beta (lo - hi) / (2 x); then the roll-in of a
// reduced opening of the same height with factor beta^2
__global__ void fri_fold_k(const E4* __restrict__ cur, size_t rows, unsigned log_rows, E4 half_beta, u32 half, u32 g_inv, const E4* __restrict__ roll, E4 beta2,
                           E4* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  u32 gp = bb_pow(g_inv, bitrev_dev(i, log_rows));
  E4 pw = e4_mul_base(half_beta, gp);
  E4 lo = cur[2 * i], hi = cur[2 * i + 1];
  E4 a = pw, b = e4_neg(pw);
  a.c[0] = bb_add(a.c[0], half);
  b.c[0] = bb_add(b.c[0], half);
  E4 r = e4_add(e4_mul(a, lo), e4_mul(b, hi));
  if (roll) r = e4_add(r, e4_mul(beta2, roll[i]));
  out[i] = r;
}
void bb_fri_fold(Ctx& ctx, const E4* cur, size_t rows, E4 beta, const E4* roll_in, E4* out) {
  unsigned lr = log2_host(rows);
  u32 half = bb_inv(bb_to_monty(2));
  fri_fold_k<<<blocks_for(rows, 256), 256, 0, ctx.stream>>>(cur, rows, lr, e4_mul_base(beta, half), half, bb_inv(bb_two_adic_generator(lr + 1)), roll_in,
                                                            e4_square(beta), out);
}

// ---- the commit-phase transcript on the device (proof-of-work bits = 0): one 16-lane group replays the duplex challenger -
// observe the round's cap, sample beta - so that the FRI rounds queue up without a host round trip; the host replays
// the same steps afterwards on its own challenger and rejects any divergence.
// one 16-lane group: observe `n_words` words of the cap (lane-visible memory), sample beta; updates *ch, fills *out
__device__ __forceinline__ void duplex_round(DevChallenger* ch, const u32* cw, u32 n_words, const Poseidon2& perm, u32 half, FriBeta* out, int l) {
  u32 s = ch->state[l];
  u32 pend = l < 8 ? ch->input[l] : 0;
  u32 n_in = ch->n_in, n_out = ch->n_out;
  auto duplex = [&]() {
    if ((u32)l < n_in) s = pend;
    n_in = 0;
    s = coop_poseidon2(perm, s, l);
    n_out = 8;
  };
  for (u32 k = 0; k < n_words; k++) {  // observe: clears the output buffer, queues, absorbs at 8
    n_out = 0;
    u32 v = cw[k];
    if ((u32)l == n_in) pend = v;
    n_in++;
    if (n_in == 8) duplex();
  }
  E4 beta;
#pragma unroll
  for (int c = 0; c < 4; c++) {  // sample_algebra_element: four pops from the back of the output buffer
    if (n_in > 0 || n_out == 0) duplex();
    beta.c[c] = (u32)__shfl((int)s, (int)(n_out - 1), 16);
    n_out--;
  }
  ch->state[l] = s;
  if (l < 8) ch->input[l] = pend;
  if (l == 0) {
    ch->n_in = n_in;
    ch->n_out = n_out;
    out->beta = beta;
    out->half_beta = e4_mul_base(beta, half);
    out->beta2 = e4_square(beta);
  }
}
__global__ __launch_bounds__(64) void fri_challenge_k(DevChallenger* ch, const Digest8* __restrict__ cap, u32 n_cap, const Poseidon2* __restrict__ perm,
                                                      u32 half, FriBeta* __restrict__ out) {
  if (threadIdx.x >= 16) return;  // one DPP row does the work: lane l holds state word l and queued input l
  duplex_round(ch, (const u32*)cap, n_cap * 8, *perm, half, out, threadIdx.x & 15);
}
void bb_fri_challenge(Ctx& ctx, DevChallenger* d_ch, const Digest8* d_cap, size_t n_cap, const Poseidon2* d_perm, FriBeta* d_out) {
  fri_challenge_k<<<1, 64, 0, ctx.stream>>>(d_ch, d_cap, (u32)n_cap, d_perm, bb_inv(bb_to_monty(2)), d_out);
}
// (squarings: the fold uses beta^(2^squarings) - a later step of a round of arity above 2)
__global__ void fri_fold_dev_k(const E4* __restrict__ cur, size_t rows, unsigned log_rows, const FriBeta* __restrict__ fb, u32 half, u32 g_inv,
                               const E4* __restrict__ roll, E4* __restrict__ out, u32 squarings) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows) return;
  E4 half_beta = fb->half_beta, beta2 = fb->beta2;
  if (squarings) {
    E4 bt = fb->beta;
    for (u32 q = 0; q < squarings; q++) bt = e4_square(bt);
    half_beta = e4_mul_base(bt, half);
    beta2 = e4_square(bt);
  }
  u32 gp = bb_pow(g_inv, bitrev_dev(i, log_rows));
  E4 pw = e4_mul_base(half_beta, gp);
  E4 lo = cur[2 * i], hi = cur[2 * i + 1];
  E4 a = pw, b = e4_neg(pw);
  a.c[0] = bb_add(a.c[0], half);
  b.c[0] = bb_add(b.c[0], half);
  E4 r = e4_add(e4_mul(a, lo), e4_mul(b, hi));
  if (roll) r = e4_add(r, e4_mul(beta2, roll[i]));
  out[i] = r;
}
// ---- every level above a layer of `len` digests in ONE launch, the BabyBear / Poseidon2 counterpart of hash.hip::subtree_k:
// a workgroup (64 sixteen-lane groups, one permutation each) owns `sub` children - a sub-tree held in LDS -, publishes its
// root write-through, draws a ticket, and the last workgroup to arrive reads all roots and finishes the tree
// (MI355X_MICROARCH.md, inter-workgroup visibility: `sc1` stores, drained, agent-scope counter, `sc1` loads).
// CH: that workgroup continues with the round's DuplexChallenger step (no proof of work: observe the root, sample beta).
// FOLD: the children are the leaf digests of a FRI layer produced here: the previous layer is folded with the beta the
// previous round left on the device, the folded layer is written and its rows hashed - a commit-phase round per launch.
struct SubtreeArgs {
  Digest8* level[24];  // level[0] = the children (len digests), level[k] = len >> k digests
  u32 len, sub;        // sub = children per workgroup (2 <= sub <= 1024, a power of two); len / sub workgroups (<= 1024)
  u32* counter;
  DevChallenger* ch;
  FriBeta* beta_out;
  u32 half;
  u32 log_rows;        // FOLD: log2 of the folded layer's length (= 2 * len)
  u32 g_inv;
  const E4* cur;
  const E4* roll;
  E4* out;
  const FriBeta* prev;
};
template <bool CH, bool FOLD>
__global__ __launch_bounds__(1024) void subtree_k(SubtreeArgs a, const Poseidon2* __restrict__ perm) {
  __shared__ __attribute__((aligned(16))) u32 sh_a[1024 * 8];  // the children of the current level ...
  __shared__ __attribute__((aligned(16))) u32 sh_b[512 * 8];   // ... and its nodes (the two swap roles level by level)
  __shared__ u32 s_last;
  u32* sh = sh_a;
  u32* sh_next = sh_b;
  const u32 t = threadIdx.x, nb = gridDim.x;
  const int l = t & 15;
  const u32 group = t >> 4;  // 64 groups
  u32 b = blockIdx.x;
  // ---- the children of this workgroup into LDS
  if (FOLD) {
    const FriBeta fb = *a.prev;
    for (u32 j = group; j < a.sub; j += 64) {
      const size_t c = size_t(b) * a.sub + j;  // leaf c = row (out[2c], out[2c + 1]) of the folded layer
      if (l < 2) {
        const size_t i = 2 * c + l;
        const u32 gp = bb_pow(a.g_inv, bitrev_dev(i, a.log_rows));
        const E4 pw = e4_mul_base(fb.half_beta, gp);
        const E4 lo = a.cur[2 * i], hi = a.cur[2 * i + 1];
        E4 x = pw, y = e4_neg(pw);
        x.c[0] = bb_add(x.c[0], a.half);
        y.c[0] = bb_add(y.c[0], a.half);
        E4 r = e4_add(e4_mul(x, lo), e4_mul(y, hi));
        if (a.roll) r = e4_add(r, e4_mul(fb.beta2, a.roll[i]));
        a.out[i] = r;
#pragma unroll
        for (int d = 0; d < 4; d++) sh[j * 8 + 4 * l + d] = r.c[d];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      u32 s = l < 8 ? sh[j * 8 + l] : 0;
      s = coop_poseidon2(*perm, s, l);
      if (l < 8) {
        sh[j * 8 + l] = s;
        a.level[0][c].w[l] = s;
      }
    }
  } else {
    const u32* src = (const u32*)(a.level[0] + size_t(b) * a.sub);
    for (u32 i = t; i < a.sub * 8; i += 1024) sh[i] = src[i];
  }
  __syncthreads();
  u32 n = a.sub >> 1, lvl = 1;
#pragma unroll 1
  for (int phase = 0; phase < 2; phase++) {
    if (phase == 1) {
      if (nb == 1) break;
      u32* roots = (u32*)a.level[lvl - 1];  // the layer of nb digests this phase just completed
      if (t < 8) __hip_atomic_store(roots + size_t(b) * 8 + t, sh[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) {
        const u32 ticket = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket == nb - 1 ? 1u : 0u;
      }
      __syncthreads();
      if (!s_last) return;
      sh = sh_a;  // room for 1024 roots
      sh_next = sh_b;
      for (u32 i = t; i < nb * 8; i += 1024) sh[i] = __hip_atomic_load(roots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == 0) __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      b = 0;
      n = nb >> 1;
    }
#pragma unroll 1
    for (; n >= 1; n >>= 1, lvl++) {
#pragma unroll 1
      for (u32 node = group; node < n; node += 64) {  // n <= 512
        u32 s = sh[node * 16 + l];  // left || right are adjacent digests
        s = coop_poseidon2(*perm, s, l);
        if (l < 8) {
          sh_next[node * 8 + l] = s;
          a.level[lvl][size_t(b) * n + node].w[l] = s;
        }
      }
      __syncthreads();
      u32* tmp = sh;
      sh = sh_next;
      sh_next = tmp;
    }
  }
  if (CH && t < 16) duplex_round(a.ch, sh, 8, *perm, a.half, a.beta_out, l);
}

static u32 subtree_children_per_group(size_t len) {
  // one workgroup up to 1024 children; above that about 256 workgroups (one per CU), each owning 2 .. 1024 children
  if (len <= 1024) return (u32)len;
  size_t sub = len / 256;
  if (sub < 2) sub = 2;
  if (sub > 1024) sub = 1024;
  return (u32)sub;
}
// allocates the layers above t's last one and fills the kernel's level table
static void subtree_levels(Ctx& ctx, BTree& t, SubtreeArgs& a) {
  const size_t len = t.sizes.back();
  if (len < 2 || len > (size_t(1) << 20)) throw std::runtime_error("subtree: layer length out of range");
  memset(&a, 0, sizeof(a));
  a.level[0] = t.layers.back().p;
  int k = 1;
  for (size_t n = len / 2; n >= 1; n /= 2) {
    t.layers.emplace_back(ctx, n);
    t.sizes.push_back(n);
    a.level[k++] = t.layers.back().p;
    if (n == 1) break;
  }
  a.len = (u32)len;
  a.sub = subtree_children_per_group(len);
  a.counter = tree_counter_slot(ctx);
  a.half = bb_inv(bb_to_monty(2));
}
static void launch_subtree(Ctx& ctx, const Poseidon2* d_perm, BTree& t, const FriRoundCh* chp) {
  SubtreeArgs a;
  const size_t len = t.sizes.back();
  subtree_levels(ctx, t, a);
  if (chp && t.cap_height != 0) throw std::runtime_error("subtree: the fused challenger step needs a root-only commitment");
  ProfScope prof(ctx, msamd::K_COMPRESS, 96.0 * double(len), double(len));
  if (chp) {
    a.ch = chp->ch;
    a.beta_out = chp->beta_out;
    subtree_k<true, false><<<(unsigned)(len / a.sub), 1024, 0, ctx.stream>>>(a, d_perm);
  } else {
    subtree_k<false, false><<<(unsigned)(len / a.sub), 1024, 0, ctx.stream>>>(a, d_perm);
  }
}
bool bb_fri_round_fusable(size_t rows, unsigned cap_height) {
  return rows >= 4 && rows / 2 <= subtree_max() && cap_height == 0 && !getenv("MSBB_NO_SUBTREE") && !getenv("MSBB_NO_FRI_FUSED");
}
// One commit-phase round in one launch: fold `cur` (2 * rows elements) with the beta of `prev` into `out` (rows elements),
// hash the rows / 2 leaves of the folded layer, build its tree `t` and run the challenger step on the root.
void bb_fri_round_fused(Ctx& ctx, const Poseidon2* d_perm, const E4* cur, size_t rows, const FriBeta* prev, const E4* roll_in, E4* out,
                        BTree& t, DevChallenger* d_ch, FriBeta* d_beta_out) {
  const size_t leaves = rows / 2;
  t = BTree();
  t.cap_height = 0;
  t.layers.emplace_back(ctx, leaves);
  t.sizes.push_back(leaves);
  SubtreeArgs a;
  subtree_levels(ctx, t, a);
  const unsigned lr = log2_host(rows);
  a.ch = d_ch;
  a.beta_out = d_beta_out;
  a.log_rows = lr;
  a.g_inv = bb_inv(bb_two_adic_generator(lr + 1));
  a.cur = cur;
  a.roll = roll_in;
  a.out = out;
  a.prev = prev;
  ProfScope prof(ctx, msamd::K_FRI_FOLD, 48.0 * double(rows) + 64.0 * double(leaves), 2.0 * double(leaves));
  subtree_k<true, true><<<(unsigned)(leaves / a.sub), 1024, 0, ctx.stream>>>(a, d_perm);
}

void bb_fri_fold_dev(Ctx& ctx, const E4* cur, size_t rows, const FriBeta* d_beta, const E4* roll_in, E4* out, unsigned squarings) {
  unsigned lr = log2_host(rows);
  fri_fold_dev_k<<<blocks_for(rows, 256), 256, 0, ctx.stream>>>(cur, rows, lr, d_beta, bb_inv(bb_to_monty(2)), bb_inv(bb_two_adic_generator(lr + 1)), roll_in,
                                                                out, (u32)squarings);
}

__global__ void gather_k(const GatherSeg* __restrict__ segs, size_t nsegs, u32* __restrict__ out) {
  size_t s = blockIdx.x;
  if (s >= nsegs) return;
  GatherSeg g = segs[s];
  for (u32 k = threadIdx.x; k < g.n; k += blockDim.x) out[g.dst + k] = g.src[(size_t)k * g.stride];
}
void bb_gather(Ctx& ctx, const std::vector<GatherSeg>& segs, std::vector<u32>& out) {
  size_t total = 0;
  for (auto& s : segs) total = std::max<size_t>(total, (size_t)s.dst + s.n);
  out.assign(total, 0);
  if (segs.empty()) return;
  DBuf<GatherSeg> d_segs(ctx, segs.size());
  DBuf<u32> d_out(ctx, std::max<size_t>(total, 1));
  ctx.h2d(d_segs.p, segs.data(), segs.size() * sizeof(GatherSeg));
  gather_k<<<(unsigned)segs.size(), 64, 0, ctx.stream>>>(d_segs.p, segs.size(), d_out.p);
  ctx.d2h(out.data(), d_out.p, total * 4);
}

// element-wise field ops for known-answer tests: 0 add, 1 sub, 2 mul, 3 inverse(a), 4 ext4 mul (quads), 5 ext4 inverse
__global__ void field_op_k(int op, const u32* a, const u32* b, size_t n, u32* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (op < 4) {
    u32 x = bb_to_monty(a[i]), y = b ? bb_to_monty(b[i]) : 0;
    u32 r = op == 0 ? bb_add(x, y) : op == 1 ? bb_sub(x, y) : op == 2 ? bb_mul(x, y) : bb_inv(x);
    out[i] = bb_from_monty(r);
  } else {
    E4 x, y = e4_zero();
    for (int k = 0; k < 4; k++) x.c[k] = bb_to_monty(a[4 * i + k]);
    if (op == 4)
      for (int k = 0; k < 4; k++) y.c[k] = bb_to_monty(b[4 * i + k]);
    E4 r = op == 4 ? e4_mul(x, y) : e4_inv(x);
    for (int k = 0; k < 4; k++) out[4 * i + k] = bb_from_monty(r.c[k]);
  }
}
void bb_field_op(Ctx& ctx, int op, const u32* a, const u32* b, size_t n, u32* out) {
  size_t words = op >= 4 ? 4 * n : n;
  DBuf<u32> da(ctx, std::max<size_t>(words, 1)), db(ctx, std::max<size_t>(words, 1)), dout(ctx, std::max<size_t>(words, 1));
  if (!n) return;
  ctx.h2d(da.p, a, words * 4);
  if (b) ctx.h2d(db.p, b, words * 4);
  field_op_k<<<blocks_for(n, 256), 256, 0, ctx.stream>>>(op, da.p, b ? db.p : nullptr, n, dout.p);
  ctx.d2h(out, dout.p, words * 4);
}

}  // namespace msbb
