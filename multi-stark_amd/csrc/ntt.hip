// Batched radix-2 Goldilocks NTT / coset LDE for gfx950.
// Replaces p3 Radix2DitParallel::dft_batch and the coset_lde_batch inside TwoAdicFriPcs::commit as reached from
// /root/reference/src/prover.rs:350,419,650,716 and src/system.rs:193.
//
// Layout: column-major (one polynomial = one contiguous column); committed LDEs keep the reference's
// bit-reversed row order (src/prover.rs:685-692) so no permutation pass is ever run:
//   * ntt_dif = in-place Gentleman-Sande: natural in -> bit-reversed out
//   * ntt_dit = in-place Cooley-Tukey:   bit-reversed in -> natural out
// A size-2^L transform is a recursive four-step: strided passes of <= 8 bits over LDS tiles of 4096 elements
// (tile = M sub-transform points x T consecutive columns of the implicit M x S matrix, so every global access is
// a run of T*8 >= 128 contiguous bytes), an inter-pass twiddle w_B^{l * bitrev(h)} from a two-level table, and a
// final contiguous pass of <= 11 bits. The coset LDE is B independent size-n transforms of the coefficient
// vector scaled by (g w_N^k0)^j / n (the first log2(B) stages of the zero-padded size-Bn transform are trivial),
// with the scaling fused into the first pass' loads.
#include <cstring>

#include "msamd.h"

namespace msamd {

namespace {

__device__ __forceinline__ u64 tw_lookup(const u64* __restrict__ t0, const u64* __restrict__ t1, u32 e26) {
  return gl_mul(t1[e26 >> TW_HALF], t0[e26 & ((1u << TW_HALF) - 1)]);
}

// radix-2 stages over `k` bits of an LDS array whose transform axis has stride `T` (T = 1 for contiguous)
template <int DIT>
__device__ __forceinline__ void lds_stages(u64* sm, unsigned k, unsigned logT, const u64* __restrict__ t1) {
  if (k == 0) return;
  const unsigned T = 1u << logT;
  const unsigned nb = (1u << (k - 1)) << logT;  // butterflies
  for (unsigned s = 0; s < k; s++) {
    const unsigned loghalf = DIT ? s : (k - 1 - s);
    const unsigned half = 1u << loghalf;
    // twiddle w_{2 half}^j = W^(j << (26 - loghalf - 1)) = T1[j << (13 - loghalf - 1)]
    const unsigned tsh = TW_HALF - loghalf - 1;
    for (unsigned b = threadIdx.x; b < nb; b += blockDim.x) {
      unsigned l = b & (T - 1), bb = b >> logT;
      unsigned j = bb & (half - 1), grp = bb >> loghalf;
      unsigned i0 = (((grp << (loghalf + 1)) + j) << logT) + l;
      unsigned i1 = i0 + (half << logT);
      u64 w = t1[j << tsh];
      u64 x = sm[i0], y = sm[i1];
      if (DIT) {
        u64 t = gl_mul(w, y);
        sm[i0] = gl_add(x, t);
        sm[i1] = gl_sub(x, t);
      } else {
        sm[i0] = gl_add(x, y);
        sm[i1] = gl_mul(gl_sub(x, y), w);
      }
    }
    __syncthreads();
  }
}

template <int DIT>
__global__ __launch_bounds__(256) void ntt_contig_k(const u64* __restrict__ src, u64* __restrict__ dst, unsigned K,
                                                     unsigned logn, const u64* __restrict__ t1, unsigned src_div,
                                                     const u64* __restrict__ scale, u64 out_mul) {
  extern __shared__ u64 sm[];
  const size_t n = size_t(1) << logn;
  const size_t col = blockIdx.y, off = size_t(blockIdx.x) << K;
  const u64* s = src + (col / src_div) * n + off;
  const u64* sc = scale ? scale + (col % src_div) * n + off : nullptr;
  u64* d = dst + col * n + off;
  const unsigned m = 1u << K;
  for (unsigned i = threadIdx.x; i < m; i += blockDim.x) {
    u64 v = s[i];
    if (sc) v = gl_mul(v, sc[i]);
    sm[i] = v;
  }
  __syncthreads();
  lds_stages<DIT>(sm, K, 0, t1);
  for (unsigned i = threadIdx.x; i < m; i += blockDim.x) {
    u64 v = sm[i];
    if (out_mul != 1) v = gl_mul(v, out_mul);
    d[i] = v;
  }
}

template <int DIT>
__global__ __launch_bounds__(256) void ntt_strided_k(const u64* __restrict__ src, u64* __restrict__ dst, unsigned k,
                                                      unsigned logS, unsigned logT, unsigned logn,
                                                      const u64* __restrict__ t0, const u64* __restrict__ t1,
                                                      unsigned src_div, const u64* __restrict__ scale, u64 out_mul) {
  extern __shared__ u64 sm[];
  const size_t n = size_t(1) << logn;
  const unsigned logB = k + logS;
  const unsigned tiles = 1u << (logS - logT);
  const size_t col = blockIdx.y;
  const unsigned tile = blockIdx.x & (tiles - 1);
  const size_t blk = blockIdx.x >> (logS - logT);
  const u32 l0 = tile << logT;
  const size_t base = blk << logB;
  const u64* s = src + (col / src_div) * n + base;
  const u64* sc = scale ? scale + (col % src_div) * n + base : nullptr;
  u64* d = dst + col * n + base;
  const unsigned total = 1u << (k + logT);
  const unsigned T = 1u << logT;
  const unsigned esh = TW_LOG - logB;
  for (unsigned idx = threadIdx.x; idx < total; idx += blockDim.x) {
    unsigned h = idx >> logT, l = idx & (T - 1);
    size_t pos = (size_t(h) << logS) + l0 + l;
    u64 v = s[pos];
    if (sc) v = gl_mul(v, sc[pos]);
    if (DIT) {
      u32 e = (l0 + l) * bitrev32(h, k);
      v = gl_mul(v, tw_lookup(t0, t1, e << esh));
    }
    sm[idx] = v;
  }
  __syncthreads();
  lds_stages<DIT>(sm, k, logT, t1);
  for (unsigned idx = threadIdx.x; idx < total; idx += blockDim.x) {
    unsigned h = idx >> logT, l = idx & (T - 1);
    size_t pos = (size_t(h) << logS) + l0 + l;
    u64 v = sm[idx];
    if (!DIT) {
      u32 e = (l0 + l) * bitrev32(h, k);
      v = gl_mul(v, tw_lookup(t0, t1, e << esh));
    }
    if (out_mul != 1) v = gl_mul(v, out_mul);
    d[pos] = v;
  }
}


// ---------------------------------------------------------------------------------------------------------
// Register radix-16 kernels. A thread owns 16 elements whose positions differ in 4 index bits, runs those 4
// radix-2 stages in registers (32 butterflies, 15 twiddle loads from the compact per-order tables) and trades
// elements with the rest of the 256-thread workgroup through LDS between rounds. A 4096-element tile therefore
// needs 3 rounds for a 12-bit contiguous pass and 2 rounds for an 8-bit strided pass, instead of 12 / 8
// LDS round trips with a barrier each. All global accesses are coalesced runs of >= 128 bytes.

// x[j] sits at position lo + (j << shift) of a block of 16 << shift positions; runs the stages over bits
// [shift, shift + 4). Root of the stage with half-size `half` (in j units) has order 2 * half << shift.
template <bool DIT>
__device__ __forceinline__ void reg_stages16(u64 (&x)[16], u32 lo, unsigned shift, const u64* __restrict__ twc) {
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int loghalf = DIT ? s : 3 - s;
    const int half = 1 << loghalf;
    const unsigned r = loghalf + 1 + shift;                  // root order 2^r
    const u64* tab = twc + ((1u << (r - 1)) - 1) + lo;        // table r, entry (jl << shift) + lo
#pragma unroll
    for (int jl = 0; jl < half; jl++) {
      const u64 w = tab[jl << shift];
#pragma unroll
      for (int g = 0; g < 8 / half; g++) {
        const int j0 = g * 2 * half + jl, j1 = j0 + half;
        u64 a = x[j0], b = x[j1];
        if (DIT) {
          u64 t = gl_mul(w, b);
          x[j0] = gl_add(a, t);
          x[j1] = gl_sub(a, t);
        } else {
          x[j0] = gl_add(a, b);
          x[j1] = gl_mul(gl_sub(a, b), w);
        }
      }
    }
  }
}


// The round over the lowest 4 position bits (lo = 0, shift = 0): all its twiddles are powers of w_16 = 2^156
// (inverse: 2^36), i.e. signed powers of two — 15 of the 32 butterflies have twiddle 1 and the other 17 are shifts.
template <bool DIT, bool INV>
__device__ __forceinline__ void reg_stages16_uniform(u64 (&x)[16]) {
  constexpr unsigned E16 = INV ? 36u : 156u;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int loghalf = DIT ? s : 3 - s;
    const int half = 1 << loghalf;
#pragma unroll
    for (int jl = 0; jl < half; jl++) {
      // twiddle w_{2 half}^jl = w_16^(jl * 8 / half) = 2^k (mod 192), sign folded into the butterfly
      const unsigned kk = (E16 * (unsigned)(jl * (8 / half))) % 192u;
      const bool neg = kk >= 96;
      const unsigned k = neg ? kk - 96 : kk;
#pragma unroll
      for (int g = 0; g < 8 / half; g++) {
        const int j0 = g * 2 * half + jl, j1 = j0 + half;
        u64 a = x[j0], b = x[j1];
        if (DIT) {
          u64 t = gl_mul_2exp(b, k);
          x[j0] = neg ? gl_sub(a, t) : gl_add(a, t);
          x[j1] = neg ? gl_add(a, t) : gl_sub(a, t);
        } else {
          x[j0] = gl_add(a, b);
          x[j1] = gl_mul_2exp(neg ? gl_sub(b, a) : gl_sub(a, b), k);
        }
      }
    }
  }
}

// Same 4 stages as reg_stages16, as two radix-4 layers: a radix-4 butterfly needs three general multiplications
// (by w, w^2, w^3) and one multiplication by w_4 = 2^48 (a shift) where two radix-2 stages need four.
// tw1 = compact tables (w^i), tw3 = cube tables (w^(3i)); w_4^-1 = -2^48 for the inverse transform.
template <bool DIT, bool INV>
__device__ __forceinline__ void reg_stages16_r4(u64 (&x)[16], u32 lo, unsigned shift, const u64* __restrict__ tw1,
                                                const u64* __restrict__ tw3) {
  // multiply by the 4th root used by this direction: forward i = 2^48, inverse i^-1 = -2^48
  auto mul_i = [](u64 pos, u64 neg) -> u64 { return INV ? gl_mul_2exp(gl_sub(neg, pos), 48) : gl_mul_2exp(gl_sub(pos, neg), 48); };
#pragma unroll
  for (int layer = 0; layer < 2; layer++) {
    const int q = DIT ? (layer == 0 ? 1 : 4) : (layer == 0 ? 4 : 1);  // quarter size in j units
    const unsigned r = (q == 4 ? 4u : 2u) + shift;                     // butterfly root order 2^r = 4 q << shift
    const u64* t1 = tw1 + ((1u << (r - 1)) - 1) + lo;                  // w   = w_{2^r}^e,      e = (jl << shift) + lo
    const u64* t2 = tw1 + ((1u << (r - 2)) - 1) + lo;                  // w^2 = w_{2^(r-1)}^e
    const u64* t3 = tw3 + ((1u << (r - 2)) - 1) + lo;                  // w^3
#pragma unroll
    for (int jl = 0; jl < q; jl++) {
      const u64 w1 = t1[jl << shift], w2 = t2[jl << shift], w3 = t3[jl << shift];
#pragma unroll
      for (int g = 0; g < 4 / q; g++) {
        const int j0 = g * 4 * q + jl;
        u64 x0 = x[j0], x1 = x[j0 + q], x2 = x[j0 + 2 * q], x3 = x[j0 + 3 * q];
        if (!DIT) {
          u64 s02 = gl_add(x0, x2), s13 = gl_add(x1, x3), d02 = gl_sub(x0, x2);
          u64 id13 = mul_i(x1, x3);  // i (x1 - x3)
          x[j0] = gl_add(s02, s13);
          x[j0 + q] = gl_mul(gl_sub(s02, s13), w2);
          x[j0 + 2 * q] = gl_mul(gl_add(d02, id13), w1);
          x[j0 + 3 * q] = gl_mul(gl_sub(d02, id13), w3);
        } else {
          u64 t1v = gl_mul(w2, x1), t2v = gl_mul(w1, x2), t3v = gl_mul(w3, x3);
          u64 a = gl_add(x0, t1v), b = gl_sub(x0, t1v), c = gl_add(t2v, t3v);
          u64 id = mul_i(t2v, t3v);  // i (t2 - t3)
          x[j0] = gl_add(a, c);
          x[j0 + 2 * q] = gl_sub(a, c);
          x[j0 + q] = gl_add(b, id);
          x[j0 + 3 * q] = gl_sub(b, id);
        }
      }
    }
  }
}

// The same four stages once more, as "16-point network with power-of-two twiddles, then one general multiplication per
// value": the stage twiddle w_{2 half S}^{jl S + lo} factors into the uniform w_{2 half}^{jl} (a shift) and a per-thread
// rho_s = w_{2 half S}^{lo}; along the network the rho factors of a value collect to rho^{rev4(j)}, rho = w_{16 S}^{lo}
// (DIF: applied after the network; DIT: before it). 15 general multiplications per 16 values instead of 24.
// twf = full per-order tables (every exponent below the order).
#ifndef MSAMD_R16TW
#define MSAMD_R16TW 1
#endif
template <bool DIT, bool INV>
__device__ __forceinline__ void reg_stages16_tw(u64 (&x)[16], u32 lo, unsigned shift, const u64* __restrict__ twf) {
  const u64* tab = twf + ((1u << (4 + shift)) - 1);
  u64 w[16];
#pragma unroll
  for (int j = 1; j < 16; j++) {
    const int c = ((j & 1) << 3) | ((j & 2) << 1) | ((j & 4) >> 1) | ((j & 8) >> 3);
    w[j] = tab[(u32)c * lo];
  }
  if (DIT) {
#pragma unroll
    for (int j = 1; j < 16; j++) x[j] = gl_mul(x[j], w[j]);
    reg_stages16_uniform<true, INV>(x);
  } else {
    reg_stages16_uniform<false, INV>(x);
#pragma unroll
    for (int j = 1; j < 16; j++) x[j] = gl_mul(x[j], w[j]);
  }
}
// general round: either form (twx = cube tables for the radix-4 form, full tables for the radix-16 form)
template <bool DIT, bool INV>
__device__ __forceinline__ void reg_round16(u64 (&x)[16], u32 lo, unsigned shift, const u64* __restrict__ twc, const u64* __restrict__ twx) {
#if MSAMD_R16TW
  (void)twc;
  reg_stages16_tw<DIT, INV>(x, lo, shift, twx);
#else
  reg_stages16_r4<DIT, INV>(x, lo, shift, twc, twx);
#endif
}

__device__ __forceinline__ u32 pad_hi(u32 e) { return e + ((e >> 8) << 4); }  // 16 spare slots per 256
__device__ __forceinline__ u32 pad_lo(u32 e) { return e + (e >> 4); }         // 1 spare slot per 16
constexpr int NTT12_LDS = 4096 + 256 + 16;
// half a tile of the strided pass (256 rows x 8): 8 spare slots per 128, so that the four row groups of a wave's transposed
// access start 16 banks apart
__device__ __forceinline__ u32 pad_half(u32 e) { return e + ((e >> 7) << 3); }
constexpr int NTT8S_LDS = 2048 + 128 + 8;

// 12-bit contiguous pass over one 4096-element tile (bits 11..0 of the position inside the tile).
template <bool DIT, bool INV>
// (tile_mul, tile_add: workgroup x takes tile x * tile_mul + tile_add - 1, 0 for every tile; ntt_dit_first_pass_part below)
__global__ __launch_bounds__(256) void ntt12_k(const u64* __restrict__ src, u64* __restrict__ dst, unsigned logn,
                                               const u64* __restrict__ twc, const u64* __restrict__ twc3, unsigned src_div,
                                               const u64* __restrict__ scale, u64 out_mul, u32 tile_mul, u32 tile_add) {
  __shared__ u64 sm[NTT12_LDS];
  const size_t n = size_t(1) << logn;
  const size_t col = blockIdx.y, off = (size_t(blockIdx.x) * tile_mul + tile_add) << 12;
  const u64* s = src + (col / src_div) * n + off;
  const u64* sc = scale ? scale + (col % src_div) * n + off : nullptr;
  u64* d = dst + col * n + off;
  const u32 t = threadIdx.x, a = t >> 4, b = t & 15;
  u64 x[16];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    u64 v = s[t + 256 * j];
    if (sc) v = gl_mul(v, sc[t + 256 * j]);
    x[j] = v;
  }
  if (!DIT) {
    reg_round16<false, INV>(x, t, 8, twc, twc3);  // bits 11..8
#pragma unroll
    for (int j = 0; j < 16; j++) sm[pad_hi(t + 256 * j)] = x[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = sm[pad_hi(a * 256 + 16 * j + b)];
    reg_round16<false, INV>(x, b, 4, twc, twc3);  // bits 7..4
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) sm[pad_lo(a * 256 + 16 * j + b)] = x[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = sm[pad_lo(16 * t + j)];
    reg_stages16_uniform<false, INV>(x);  // bits 3..0
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) sm[pad_lo(16 * t + j)] = x[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = sm[pad_lo(t + 256 * j)];
  } else {
#pragma unroll
    for (int j = 0; j < 16; j++) sm[pad_lo(t + 256 * j)] = x[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = sm[pad_lo(16 * t + j)];
    reg_stages16_uniform<true, INV>(x);  // bits 0..3
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) sm[pad_lo(16 * t + j)] = x[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = sm[pad_lo(a * 256 + 16 * j + b)];
    reg_round16<true, INV>(x, b, 4, twc, twc3);  // bits 4..7
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) sm[pad_hi(a * 256 + 16 * j + b)] = x[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = sm[pad_hi(t + 256 * j)];
    reg_round16<true, INV>(x, t, 8, twc, twc3);  // bits 8..11
  }
#pragma unroll
  for (int j = 0; j < 16; j++) {
    u64 v = x[j];
    if (out_mul != 1) v = gl_mul(v, out_mul);
    d[t + 256 * j] = v;
  }
}

// ---- strided passes of 1 .. 6 bits entirely in registers (heights that are not 12 + 8 k bits: 2^13 .. 2^19 rows, the last
// strided pass of 2^26). Every root of unity of order <= 64 is a power of two (w_64 = 2^39, w_32 = 2^78, w_16 = 2^156, ... ;
// 2 has order 192), so a 2^K-point sub-transform costs additions and SHIFTS only: a thread owns the 2^K values of one position
// l (stride S apart; the lanes of a wave own consecutive l: every access is a 512-byte run), runs the K radix-2 stages in
// registers without any exchange, and multiplies by the inter-pass twiddle w_B^(l bitrev_K(h)) - the same arithmetic as
// ntt_strided_k's K barrier-separated LDS stages, which this replaces (37 -> us per launch on the reference's BLAKE3 system).
template <int K, bool DIT, bool INV>
__device__ __forceinline__ void reg_stages_small(u64 (&x)[1 << K]) {
  // exponent e with w_{2^K} = 2^e for the forward transform (the inverse root is 2^(192 - e))
  constexpr unsigned EF = K == 1 ? 96u : K == 2 ? 48u : K == 3 ? 120u : K == 4 ? 156u : K == 5 ? 78u : 39u;
  constexpr unsigned E = INV ? 192u - EF : EF;
  constexpr int N = 1 << K;
#pragma unroll
  for (int s = 0; s < K; s++) {
    const int loghalf = DIT ? s : K - 1 - s;
    const int half = 1 << loghalf;
#pragma unroll
    for (int jl = 0; jl < half; jl++) {
      // twiddle w_{2 half}^jl = w_{2^K}^(jl 2^(K-1) / half), a signed power of two
      const unsigned kk = (E * (unsigned)(jl * ((N / 2) / half))) % 192u;
      const bool neg = kk >= 96;
      const unsigned k = neg ? kk - 96 : kk;
#pragma unroll
      for (int g = 0; g < (N / 2) / half; g++) {
        const int j0 = g * 2 * half + jl, j1 = j0 + half;
        const u64 a = x[j0], b = x[j1];
        if (DIT) {
          const u64 t = gl_mul_2exp(b, k);
          x[j0] = neg ? gl_sub(a, t) : gl_add(a, t);
          x[j1] = neg ? gl_add(a, t) : gl_sub(a, t);
        } else {
          x[j0] = gl_add(a, b);
          x[j1] = gl_mul_2exp(neg ? gl_sub(b, a) : gl_sub(a, b), k);
        }
      }
    }
  }
}

template <int K, bool DIT, bool INV>
__global__ __launch_bounds__(256) void ntt_small_strided_k(const u64* __restrict__ src, u64* __restrict__ dst, unsigned logS, unsigned logn,
                                                           const u64* __restrict__ t0, const u64* __restrict__ t1, unsigned src_div,
                                                           const u64* __restrict__ scale, u64 out_mul) {
  constexpr int N = 1 << K;
  const size_t n = size_t(1) << logn;
  const unsigned logB = K + logS;
  const size_t gid = blockIdx.x * size_t(blockDim.x) + threadIdx.x;  // (block of 2^logB positions, l)
  if (gid >= (n >> K)) return;
  const size_t col = blockIdx.y;
  const u32 lg = (u32)(gid & ((size_t(1) << logS) - 1));
  const size_t base = (gid >> logS) << logB;
  const u64* s = src + (col / src_div) * n + base;
  const u64* sc = scale ? scale + (col % src_div) * n + base : nullptr;
  u64* d = dst + col * n + base;
  const unsigned esh = TW_LOG - logB;
  u64 x[N];
#pragma unroll
  for (int h = 0; h < N; h++) x[h] = s[(size_t(h) << logS) + lg];
  if (sc) {
#pragma unroll
    for (int h = 0; h < N; h++) x[h] = gl_mul(x[h], sc[(size_t(h) << logS) + lg]);
  }
  if (DIT) {
#pragma unroll
    for (int h = 1; h < N; h++) x[h] = gl_mul(x[h], tw_lookup(t0, t1, (lg * bitrev32((u32)h, K)) << esh));
  }
  reg_stages_small<K, DIT, INV>(x);
#pragma unroll
  for (int h = 0; h < N; h++) {
    u64 v = x[h];
    if (!DIT && h) v = gl_mul(v, tw_lookup(t0, t1, (lg * bitrev32((u32)h, K)) << esh));
    if (out_mul != 1) v = gl_mul(v, out_mul);
    d[(size_t(h) << logS) + lg] = v;
  }
}

// 8-bit strided pass: sub-transforms of 256 points at stride S = 2^logS inside blocks of 2^(8 + logS); a tile is
// 256 (h) x 16 (l) with l contiguous in memory. Includes the four-step inter-pass twiddle w_B^{l * bitrev8(h)}.
// The forward branch issues its 16 raw loads first and the 16 scale loads + multiplications after them: the strided pass
// is the one kernel where waves still park on HBM latency (profiles/r01_pmc_sq.txt), and interleaving load / scale load /
// multiply per element kept fewer requests in flight (-9 % on the launch). Taking two tiles per workgroup and prefetching
// the second was tried as well: no gain, it halves the occupancy.
// (six workgroups per CU for the inverse direction: 80 registers, five of them spilled - measured 153 -> 144 us; the forward
// direction keeps its 108: squeezed to 96 it spills twelve and loses 5 %)
template <bool DIT, bool INV>
__global__ __launch_bounds__(256, DIT ? 6 : 4) void ntt8s_k(const u64* __restrict__ src, u64* __restrict__ dst, unsigned logS, unsigned logn,
                                               const u64* __restrict__ twc, const u64* __restrict__ twc3,
                                               const u64* __restrict__ t0, const u64* __restrict__ t1,
                                               const u64* __restrict__ ttab, unsigned src_div, const u64* __restrict__ scale,
                                               u64 out_mul, u32 gx, u32 ncols) {
  // The one exchange of this pass only moves values between the 16 threads that share l, so the two halves of the tile
  // (l < 8, l >= 8) go through the SAME 17 KB one after the other: twice the workgroups fit a CU's LDS (the 35 KB of the whole
  // tile capped the pass at four waves per SIMD, and it waits on its strided loads for a quarter of its time)
  __shared__ u64 sm[NTT8S_LDS];
  const size_t n = size_t(1) << logn;
  const unsigned logB = 8 + logS;
  const unsigned tiles = 1u << (logS - 4);
  // Tile -> workgroup mapping. The coset scale and the inter-pass twiddle of a tile are the same for every column, and
  // together they are as many bytes as the tile itself: with one column per grid row they were fetched again for each of
  // the 56 / 104 columns (FETCH_SIZE 1.40 x the algorithmic bytes, profiles/r01_traffic.json). Workgroups are dispatched
  // round-robin over the 8 XCDs, each with its own L2: XCD x now walks through ALL columns of one tile before it moves
  // to the next, so a table tile enters an L2 once and serves every column from there.
  size_t col;
  u32 bx;
  {
    const size_t id = blockIdx.x;
    if ((gx & 7u) == 0) {
      const size_t q = id >> 3;
      col = q % ncols;
      bx = (u32)((q / ncols) * 8 + (id & 7));
    } else {
      col = id % ncols;
      bx = (u32)(id / ncols);
    }
  }
  const u32 t = threadIdx.x, hq = t >> 4, l = t & 15;  // hq: low 4 bits of h in the strided round, high 4 in the other
  const unsigned esh = TW_LOG - logB;
  const u32 rev_hq = bitrev32(hq, 4);
  u64 x[16];
  if (!DIT) {
    const u32 lg = ((bx & (tiles - 1)) << 4) + l;
    const size_t base = size_t(bx >> (logS - 4)) << logB;
    const u64* s = src + (col / src_div) * n + base;
    const u64* sc = scale ? scale + (col % src_div) * n + base : nullptr;
    u64* d = dst + col * n + base;
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = s[(size_t(hq + 16 * j) << logS) + lg];
    if (sc) {
#pragma unroll
      for (int j = 0; j < 16; j++) x[j] = gl_mul(x[j], sc[(size_t(hq + 16 * j) << logS) + lg]);
    }
    reg_round16<false, INV>(x, hq, 4, twc, twc3);  // h bits 7..4
#pragma unroll
    for (u32 half = 0; half < 2; half++) {
      if ((l >> 3) == half) {
#pragma unroll
        for (int j = 0; j < 16; j++) sm[pad_half((hq + 16 * j) * 8 + (l & 7))] = x[j];
      }
      __syncthreads();
      if ((l >> 3) == half) {
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = sm[pad_half((hq * 16 + j) * 8 + (l & 7))];
      }
      __syncthreads();
    }
    reg_stages16_uniform<false, INV>(x);  // h bits 3..0
    u64 tw[16];  // the sixteen inter-pass twiddles are requested together, before the first product needs one
#pragma unroll
    for (int j = 0; j < 16; j++) {
      // h = hq * 16 + j, bitrev8(h) = bitrev4(j) * 16 + bitrev4(hq)
      const u32 rev = (u32)(((j & 1) << 3 | (j & 2) << 1 | (j & 4) >> 1 | (j & 8) >> 3) << 4) + rev_hq;
      tw[j] = ttab ? ttab[(size_t(hq * 16 + j) << logS) + lg] : tw_lookup(t0, t1, (lg * rev) << esh);
    }
#pragma unroll
    for (int j = 0; j < 16; j++) {
      u64 v = gl_mul(x[j], tw[j]);
      if (out_mul != 1) v = gl_mul(v, out_mul);
      d[(size_t(hq * 16 + j) << logS) + lg] = v;
    }
  } else {
    const u32 tile = bx & (tiles - 1);
    const size_t blk = bx >> (logS - 4);
    const u32 l0 = tile << 4;
    const size_t base = blk << logB;
    const u64* s = src + (col / src_div) * n + base;
    u64* d = dst + col * n + base;
    const u32 lg = l0 + l;
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = s[(size_t(hq * 16 + j) << logS) + lg];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const u32 rev = (u32)(((j & 1) << 3 | (j & 2) << 1 | (j & 4) >> 1 | (j & 8) >> 3) << 4) + rev_hq;
      x[j] = gl_mul(x[j], ttab ? ttab[(size_t(hq * 16 + j) << logS) + lg] : tw_lookup(t0, t1, (lg * rev) << esh));
    }
    reg_stages16_uniform<true, INV>(x);  // h bits 0..3
#pragma unroll
    for (u32 half = 0; half < 2; half++) {
      if ((l >> 3) == half) {
#pragma unroll
        for (int j = 0; j < 16; j++) sm[pad_half((hq * 16 + j) * 8 + (l & 7))] = x[j];
      }
      __syncthreads();
      if ((l >> 3) == half) {
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = sm[pad_half((hq + 16 * j) * 8 + (l & 7))];
      }
      __syncthreads();
    }
    reg_round16<true, INV>(x, hq, 4, twc, twc3);  // h bits 4..7
#pragma unroll
    for (int j = 0; j < 16; j++) {
      u64 v = x[j];
      if (out_mul != 1) v = gl_mul(v, out_mul);
      d[(size_t(hq + 16 * j) << logS) + lg] = v;
    }
  }
}

// inter-pass twiddle table of the 8-bit strided pass on blocks of 2^logB: tab[h * S + l] = w_B^{l * bitrev8(h)}
__global__ void ntt8s_table_k(u64* __restrict__ tab, unsigned logB, const u64* __restrict__ t0, const u64* __restrict__ t1) {
  const size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (i >= (size_t(1) << logB)) return;
  const unsigned logS = logB - 8;
  const u32 h = (u32)(i >> logS), l = (u32)(i & ((size_t(1) << logS) - 1));
  tab[i] = tw_lookup(t0, t1, (l * bitrev32(h, 8)) << (TW_LOG - logB));
}

const u64* ntt8s_table(Ctx& ctx, unsigned logB, bool inverse) {
  if (logB > 22) return nullptr;  // 32 MiB at most; larger blocks compute the factor from the two-level tables
  auto key = std::make_pair(logB + (inverse ? 100u : 0u), 0xFFu);
  auto it = ctx.lde_scales.find(key);
  if (it != ctx.lde_scales.end()) return it->second;
  size_t cnt = size_t(1) << logB;
  u64* p = nullptr;
  HIP_CHECK(hipMalloc(&p, cnt * 8));
  hipLaunchKernelGGL(ntt8s_table_k, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx.stream, p, logB,
                     inverse ? ctx.tw0i : ctx.tw0, inverse ? ctx.tw1i : ctx.tw1);
  HIP_CHECK(hipGetLastError());
  ctx.lde_scales[key] = p;
  return p;
}

struct Plan {
  unsigned K;
  std::vector<unsigned> bits;  // strided passes, top first
};
Plan make_plan(unsigned logn) {
  Plan p;
  if (logn >= 12) {
    // register kernels: 12-bit contiguous pass, 8-bit strided passes first, the remainder on the generic kernel
    p.K = 12;
    unsigned R = logn - 12;
    while (R > 8) {
      p.bits.push_back(8);
      R -= 8;
    }
    if (R) p.bits.push_back(R);
    return p;
  }
  p.K = logn;
  return p;
}

template <int DIT>
void launch_strided(Ctx& ctx, const u64* src, u64* dst, unsigned k, unsigned logB, unsigned logn, size_t ncols,
                    bool inverse, unsigned src_div, const u64* scale, u64 out_mul) {
  unsigned logS = logB - k;
  if (k == 8 && logS >= 4) {
    size_t gx8 = (size_t(1) << (logS - 4)) << (logn - logB);
    const int id = DIT ? K_NTT8S_DIT : K_NTT8S_DIF;
    hipEvent_t ev8 = ctx.prof_begin(id);
    const u64* ttab = ntt8s_table(ctx, logB, inverse);
    if (gx8 * ncols > 0x7FFFFFFFull) throw std::runtime_error("ntt: too many tiles in one launch");
    if (inverse)
      hipLaunchKernelGGL((ntt8s_k<(DIT != 0), true>), dim3((unsigned)(gx8 * ncols)), dim3(256), 0, ctx.stream, src, dst, logS, logn,
                         ctx.twci, MSAMD_R16TW ? ctx.twfi : ctx.twc3i, ctx.tw0i, ctx.tw1i, ttab, src_div, scale, out_mul, (u32)gx8,
                         (u32)ncols);
    else
      hipLaunchKernelGGL((ntt8s_k<(DIT != 0), false>), dim3((unsigned)(gx8 * ncols)), dim3(256), 0, ctx.stream, src, dst, logS, logn,
                         ctx.twc, MSAMD_R16TW ? ctx.twf : ctx.twc3, ctx.tw0, ctx.tw1, ttab, src_div, scale, out_mul, (u32)gx8,
                         (u32)ncols);
    ctx.prof_end(id, ev8, 16.0 * double(ncols) * double(size_t(1) << logn));
    return;
  }
  if (k >= 1 && k <= 6 && logS >= 8 && !getenv("MSAMD_NO_NTT_SMALL")) {
    // a sub-transform of at most 64 points: shifts and additions in registers (ntt_small_strided_k)
    const size_t threads_total = (size_t(1) << logn) >> k;
    const u64 *t0 = inverse ? ctx.tw0i : ctx.tw0, *t1 = inverse ? ctx.tw1i : ctx.tw1;
    hipEvent_t ev = ctx.prof_begin(K_NTT_STRIDED);
    // the columns ride in grid.y (at most 65535 per launch): wider matrices take several launches
    const size_t sd = src_div ? src_div : 1, per_launch = std::max<size_t>(sd, 65535 / sd * sd);  // (whole groups of src_div columns)
    if (per_launch > 65535) throw std::runtime_error("ntt: too many output columns per source column");
    for (size_t c0 = 0; c0 < ncols; c0 += per_launch) {
      const dim3 grid((unsigned)((threads_total + 255) / 256), (unsigned)std::min<size_t>(ncols - c0, per_launch));
      const u64* s = src + ((c0 / sd) << logn);
      u64* d = dst + (c0 << logn);
#define MS_SMALL(KK)                                                                                                                 \
  case KK:                                                                                                                           \
    if (inverse)                                                                                                                     \
      hipLaunchKernelGGL((ntt_small_strided_k<KK, (DIT != 0), true>), grid, dim3(256), 0, ctx.stream, s, d, logS, logn, t0, t1, src_div, scale, out_mul); \
    else                                                                                                                             \
      hipLaunchKernelGGL((ntt_small_strided_k<KK, (DIT != 0), false>), grid, dim3(256), 0, ctx.stream, s, d, logS, logn, t0, t1, src_div, scale, out_mul); \
    break;
      switch (k) {
        MS_SMALL(1) MS_SMALL(2) MS_SMALL(3) MS_SMALL(4) MS_SMALL(5) MS_SMALL(6)
      }
#undef MS_SMALL
      HIP_CHECK(hipGetLastError());
    }
    ctx.prof_end(K_NTT_STRIDED, ev, 16.0 * double(ncols) * double(size_t(1) << logn));
    return;
  }
  unsigned logT = 12 - k;
  if (logT > logS) logT = logS;
  size_t gx = (size_t(1) << (logS - logT)) << (logn - logB);
  if (ncols > 65535) throw std::runtime_error("ntt: more than 65535 columns in one generic strided pass");
  dim3 grid((unsigned)gx, (unsigned)ncols);
  size_t shmem = (size_t(8) << (k + logT));
  hipEvent_t ev = ctx.prof_begin(K_NTT_STRIDED);
  hipLaunchKernelGGL(ntt_strided_k<DIT>, grid, dim3(256), shmem, ctx.stream, src, dst, k, logS, logT, logn,
                     inverse ? ctx.tw0i : ctx.tw0, inverse ? ctx.tw1i : ctx.tw1, src_div, scale, out_mul);
  HIP_CHECK(hipGetLastError());
  ctx.prof_end(K_NTT_STRIDED, ev, 16.0 * double(ncols) * double(size_t(1) << logn));
}

template <int DIT>
void launch_contig(Ctx& ctx, const u64* src, u64* dst, unsigned K, unsigned logn, size_t ncols, bool inverse,
                   unsigned src_div, const u64* scale, u64 out_mul, u32 tile_mul = 1, u32 tile_add = 0) {
  if (tile_mul != 1 && K != 12) throw std::runtime_error("ntt: a tile subset needs the 12-bit pass");
  dim3 grid((unsigned)((size_t(1) << (logn - K)) / tile_mul), (unsigned)ncols);
  if (K == 12) {
    const int id = DIT ? K_NTT12_DIT : K_NTT12_DIF;
    hipEvent_t ev12 = ctx.prof_begin(id);
    if (inverse)
      hipLaunchKernelGGL((ntt12_k<(DIT != 0), true>), grid, dim3(256), 0, ctx.stream, src, dst, logn, ctx.twci, MSAMD_R16TW ? ctx.twfi : ctx.twc3i, src_div, scale, out_mul,
                         tile_mul, tile_add);
    else
      hipLaunchKernelGGL((ntt12_k<(DIT != 0), false>), grid, dim3(256), 0, ctx.stream, src, dst, logn, ctx.twc, MSAMD_R16TW ? ctx.twf : ctx.twc3, src_div, scale, out_mul,
                         tile_mul, tile_add);
    ctx.prof_end(id, ev12, 16.0 * double(ncols) * double(size_t(1) << logn) / tile_mul);
    return;
  }
  unsigned threads = K >= 9 ? 256 : 64;
  hipEvent_t ev = ctx.prof_begin(K_NTT_CONTIG);
  hipLaunchKernelGGL(ntt_contig_k<DIT>, grid, dim3(threads), size_t(8) << K, ctx.stream, src, dst, K, logn,
                     inverse ? ctx.tw1i : ctx.tw1, src_div, scale, out_mul);
  ctx.prof_end(K_NTT_CONTIG, ev, 16.0 * double(ncols) * double(size_t(1) << logn));
}

void check_dims(unsigned logn, size_t ncols) {
  if (logn > NTT_MAX_LOG) throw std::runtime_error("ntt: transform larger than 2^26 is not supported");
  if (ncols > 65535) throw std::runtime_error("ntt: more than 65535 columns in one batch");
}

// scale[b][j] = (g * w_N^{bitrev_B(b)})^j / n
__global__ void lde_scale_k(u64* out, unsigned logn, unsigned lb) {
  size_t n = size_t(1) << logn;
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (i >= (n << lb)) return;
  size_t b = i >> logn, j = i & (n - 1);
  u64 wN = gl_two_adic_generator(logn + lb);
  u64 s = gl_mul(GL_GEN, gl_pow(wN, bitrev32((u32)b, lb)));
  u64 ninv = gl_inv((u64)n);
  out[i] = gl_mul(gl_pow(s, j), ninv);
}

__global__ void transpose_in_k(const u64* __restrict__ in, u64* __restrict__ out, size_t h, size_t w, unsigned logh,
                               int bitrev_rows) {
  // block: 64 output rows rr0.. x all columns, staged through LDS in chunks of 64 columns
  __shared__ u64 tile[64][65];
  const size_t rr0 = size_t(blockIdx.x) * 64;
  const unsigned rows = (unsigned)((h - rr0) < 64 ? (h - rr0) : 64);
  for (size_t c0 = 0; c0 < w; c0 += 64) {
    unsigned cols = (unsigned)((w - c0) < 64 ? (w - c0) : 64);
    for (unsigned idx = threadIdx.x; idx < rows * cols; idx += blockDim.x) {
      unsigned rr = idx / cols, c = idx % cols;
      size_t r = bitrev_rows ? bitrev64(rr0 + rr, logh) : (rr0 + rr);
      tile[rr][c] = in[r * w + c0 + c];
    }
    __syncthreads();
    for (unsigned idx = threadIdx.x; idx < rows * cols; idx += blockDim.x) {
      unsigned c = idx / rows, rr = idx % rows;
      out[(c0 + c) * h + rr0 + rr] = tile[rr][c];
    }
    __syncthreads();
  }
}

// The same with rows bit-reversed, for h >= 256. A block of 64 consecutive storage rows is 64 input rows that lie h / 64
// rows apart: every read a lone row (112 bytes at the bench's width). Here a tile is 16 x 16 storage rows
// rr = (A : mid : B), A the top and B the low four bits, i.e. natural rows (rev B : rev mid : rev A): for each of the 16
// values of B the 16 values of A are CONSECUTIVE input rows (one run of 16 w words), and for each A the 16 values of B
// are consecutive storage rows (128-byte runs per column). Columns go through LDS 16 at a time.
// (blockIdx.y: a short, wide matrix - 512 rows x 2625 columns in the reference's BLAKE3 system - gives only h / 256 row blocks, so
// the columns are dealt out over a second grid dimension, cols_per_block at a time, instead of one block walking all of them)
// (sel_bits, sel_shift, sel_val: with sel_bits > 0 the grid covers only the blocks whose `mid` holds sel_val in the sel_bits bits
// from sel_shift up - transpose_in_rows_part)
__global__ __launch_bounds__(256) void transpose_in_br_k(const u64* __restrict__ in, u64* __restrict__ out, size_t h, size_t w,
                                                         unsigned logh, u32 cols_per_block, u32 sel_bits, u32 sel_shift, u32 sel_val) {
  __shared__ u64 tile[256][17];
  const u32 mid = sel_bits ? ((blockIdx.x >> sel_shift) << (sel_shift + sel_bits)) | (sel_val << sel_shift) | (blockIdx.x & ((1u << sel_shift) - 1u))
                           : blockIdx.x;
  const size_t rmid = size_t(bitrev32(mid, logh - 8)) << 4;
  const u32 t = threadIdx.x;
  const u32 lo = t & 15, hi = t >> 4;
  const size_t c_begin = size_t(blockIdx.y) * cols_per_block, c_end = min(w, c_begin + cols_per_block);
  for (size_t c0 = c_begin; c0 < c_end; c0 += 16) {
    const u32 cols = (u32)((c_end - c0) < 16 ? (c_end - c0) : 16);
    // read: thread (hi = natural low bits a, lo = column), 16 runs u = natural top bits
    if (lo < cols) {
      const u32 A = bitrev32(hi, 4);
#pragma unroll 4
      for (u32 u = 0; u < 16; u++) {
        const size_t r = (size_t(u) << (logh - 4)) | rmid | hi;
        tile[A * 16 + bitrev32(u, 4)][lo] = in[r * w + c0 + lo];
      }
    }
    __syncthreads();
    // write: thread (hi = A, lo = B), one column per step
    {
      const size_t rr = (size_t(hi) << (logh - 4)) | (size_t(mid) << 4) | lo;
      for (u32 c = 0; c < cols; c++) out[(c0 + c) * h + rr] = tile[hi * 16 + lo][c];
    }
    __syncthreads();
  }
}

__global__ void transpose_out_k(const u64* __restrict__ in, u64* __restrict__ out, size_t h, size_t w, unsigned logh,
                                int bitrev_rows) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (i >= h * w) return;
  size_t r = i / w, c = i % w;
  size_t rs = bitrev_rows ? bitrev64(r, logh) : r;
  out[i] = in[c * h + rs];
}

// quotient: DFT output S (natural order, nq x D) -> per coset block b, slice k: pre-scaled coefficients
//   lde[(k*D + c) * Bn + b*n + r] = S[c][(N - (k n + r)) mod N] * w_k * w_{Bn}^{bitrev_B(b) * r}
struct WkTab {
  u64 w[16];  // the q slice weights inside the argument block (q <= 16), so that no upload precedes the launch
};
__global__ void quotient_slice_k(const u64* __restrict__ S, u64* __restrict__ lde, unsigned logn, unsigned logq,
                                 unsigned lb, unsigned D, const u64* __restrict__ t0, const u64* __restrict__ t1,
                                 WkTab tab, const u64* __restrict__ wk /* q weights in device memory when q > 16, else null */) {
  const size_t n = size_t(1) << logn, N = n << logq, Bn = n << lb;
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;  // over r + n*(b + B*k)
  if (i >= (n << (lb + logq))) return;
  size_t r = i & (n - 1);
  unsigned b = (unsigned)((i >> logn) & ((1u << lb) - 1));
  unsigned k = (unsigned)(i >> (logn + lb));
  size_t j = (size_t(k) << logn) + r;
  size_t srcrow = (N - j) & (N - 1);
  u32 k0 = bitrev32(b, lb);
  // w_{Bn}^{k0 r}: exponent modulo Bn, scaled to the order-2^26 table
  u64 e = (u64(k0) * r) & (Bn - 1);
  u64 f = gl_mul(wk ? wk[k] : tab.w[k & 15], tw_lookup(t0, t1, (u32)(e << (TW_LOG - logn - lb))));
  for (unsigned c = 0; c < D; c++) lde[(size_t(k) * D + c) * Bn + (size_t(b) << logn) + r] = gl_mul(S[c * N + srcrow], f);
}

}  // namespace

void ntt_dif(Ctx& ctx, u64* data, unsigned logn, size_t ncols, bool inverse, const NttSrc* from, u64 out_mul) {
  check_dims(logn, ncols);
  if (ncols == 0) return;
  Plan p = make_plan(logn);
  const u64* src = from && from->src ? from->src : data;
  unsigned src_div = from ? from->src_div : 1;
  const u64* scale = from ? from->scale : nullptr;
  unsigned logB = logn;
  for (size_t i = 0; i < p.bits.size(); i++) {
    launch_strided<0>(ctx, src, data, p.bits[i], logB, logn, ncols, inverse, src_div, scale, 1);
    src = data;
    src_div = 1;
    scale = nullptr;
    logB -= p.bits[i];
  }
  launch_contig<0>(ctx, src, data, p.K, logn, ncols, inverse, src_div, scale, out_mul);
}

// The first pass of ntt_dit (4096-row tiles of the bit-reversed storage) on the tiles t with t mod 8 == part_rev only - the
// natural rows whose residue modulo 2^(logn - 12) has part = rev3(part_rev) in its top three bits, i.e. one "row group" of a
// host-resident trace that arrives in eight groups (prover.hip, HostUpload). ntt_dit(..., first_pass_done = true) does the rest.
void ntt_dit_first_pass_part(Ctx& ctx, u64* data, unsigned logn, size_t ncols, bool inverse, unsigned part_rev) {
  check_dims(logn, ncols);
  if (logn < 15 || part_rev >= 8) throw std::runtime_error("ntt: a first pass by row groups needs 2^15 rows");
  if (ncols == 0) return;
  launch_contig<1>(ctx, data, data, 12, logn, ncols, inverse, 1, nullptr, 1, 8, part_rev);
}

void ntt_dit(Ctx& ctx, u64* data, unsigned logn, size_t ncols, bool inverse, u64 out_mul, bool first_pass_done) {
  check_dims(logn, ncols);
  if (ncols == 0) return;
  Plan p = make_plan(logn);
  if (first_pass_done && p.bits.empty()) throw std::runtime_error("ntt: no pass behind the first one");
  if (!first_pass_done) launch_contig<1>(ctx, data, data, p.K, logn, ncols, inverse, 1, nullptr, p.bits.empty() ? out_mul : 1);
  unsigned logB = p.K;
  for (size_t i = p.bits.size(); i-- > 0;) {
    logB += p.bits[i];
    launch_strided<1>(ctx, data, data, p.bits[i], logB, logn, ncols, inverse, 1, nullptr, i == 0 ? out_mul : 1);
  }
}

const u64* Ctx::lde_scale(unsigned log_n, unsigned log_blowup) {
  auto key = std::make_pair(log_n, log_blowup);
  auto it = lde_scales.find(key);
  if (it != lde_scales.end()) return it->second;
  size_t cnt = size_t(1) << (log_n + log_blowup);
  u64* p = nullptr;
  HIP_CHECK(hipMalloc(&p, cnt * 8));
  hipLaunchKernelGGL(lde_scale_k, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, stream, p, log_n, log_blowup);
  HIP_CHECK(hipGetLastError());
  lde_scales[key] = p;
  return p;
}

void lde_from_coeffs(Ctx& ctx, const u64* coef, u64* lde, unsigned logn, unsigned log_blowup, size_t w) {
  NttSrc from;
  from.src = coef;
  from.src_div = 1u << log_blowup;
  from.scale = ctx.lde_scale(logn, log_blowup);
  ntt_dif(ctx, lde, logn, w << log_blowup, false, &from, 1);
}

void coset_lde(Ctx& ctx, u64* evals_bitrev, u64* lde, unsigned logn, unsigned log_blowup, size_t w, bool first_pass_done) {
  ntt_dit(ctx, evals_bitrev, logn, w, true, 1, first_pass_done);  // unscaled inverse DFT: n * coefficients, natural order
  lde_from_coeffs(ctx, evals_bitrev, lde, logn, log_blowup, w);
}

void quotient_lde(Ctx& ctx, u64* qvals_bitrev, u64* lde, unsigned logn, unsigned logq, unsigned log_blowup, size_t D) {
  // forward DFT of the quotient evaluations (bit-reversed in -> natural out), src/prover.rs:650
  unsigned logN = logn + logq;
  ntt_dit(ctx, qvals_bitrev, logN, D, false, 1);
  size_t q = size_t(1) << logq, n = size_t(1) << logn;
  // w_k = N^-1 * GENERATOR^(-k n), src/prover.rs:651-657
  std::vector<u64> wk(q);
  u64 n_inv = gl_inv((u64)(n << logq));
  u64 step = gl_inv(gl_pow(GL_GEN, (u64)n)), cur = 1;
  for (size_t k = 0; k < q; k++) {
    wk[k] = gl_mul(cur, n_inv);
    cur = gl_mul(cur, step);
  }
  WkTab tab;
  memset(&tab, 0, sizeof(tab));
  DBuf<u64> dwk;
  if (q <= 16) {
    for (size_t k = 0; k < q; k++) tab.w[k] = wk[k];
  } else {
    dwk = DBuf<u64>(ctx, q);
    ctx.h2d(dwk.p, wk.data(), q * 8);
  }
  size_t total = n << (log_blowup + logq);
  hipLaunchKernelGGL(quotient_slice_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx.stream, qvals_bitrev, lde,
                     logn, logq, log_blowup, (unsigned)D, ctx.tw0, ctx.tw1, tab, (const u64*)dwk.p);
  HIP_CHECK(hipGetLastError());
  // B independent size-n transforms per output column (q*D columns), src/prover.rs:716
  ntt_dif(ctx, lde, logn, (q * D) << log_blowup, false, nullptr, 1);
}

void transpose_in(Ctx& ctx, const u64* rowmajor, u64* colmajor, size_t h, size_t w, bool bitrev_rows) {
  if (h * w == 0) return;
  hipEvent_t ev = ctx.prof_begin(K_TRANSPOSE);
  const unsigned logh = log2_strict(h);
  if (bitrev_rows && logh >= 8 && (size_t(1) << logh) == h && !getenv("MSAMD_OLD_TRANSPOSE"))
  {
    const size_t bx = h >> 8, col_tiles = (w + 15) / 16;
    size_t by = bx < 1024 ? std::min(col_tiles, (1024 + bx - 1) / bx) : 1;  // about a thousand workgroups at least
    const size_t tiles_per_block = (col_tiles + by - 1) / by;
    by = (col_tiles + tiles_per_block - 1) / tiles_per_block;
    hipLaunchKernelGGL(transpose_in_br_k, dim3((unsigned)bx, (unsigned)by), dim3(256), 0, ctx.stream, rowmajor, colmajor, h, w, logh,
                       (u32)(tiles_per_block * 16), 0u, 0u, 0u);
  }
  else
    hipLaunchKernelGGL(transpose_in_k, dim3((unsigned)((h + 63) / 64)), dim3(256), 0, ctx.stream, rowmajor, colmajor, h, w, logh,
                       bitrev_rows ? 1 : 0);
  HIP_CHECK(hipGetLastError());
  ctx.prof_end(K_TRANSPOSE, ev, 16.0 * double(h) * double(w));
}

// transpose_in(.., bitrev_rows = true) for ONE of eight row groups: the natural rows r whose residue modulo 2^T, T = log2 h - 12,
// has `part` in its top three bits (runs of 2^(T-3) consecutive rows every 2^T). Those are the blocks of transpose_in_br_k
// whose `mid` has rev3(part) in bits 8..10, and they fill the 4096-row storage tiles t with t mod 8 == rev3(part). h >= 2^19.
void transpose_in_rows_part(Ctx& ctx, const u64* rowmajor, u64* colmajor, size_t h, size_t w, unsigned part) {
  const unsigned logh = log2_strict(h);
  if ((size_t(1) << logh) != h || logh < 19 || part >= 8) throw std::runtime_error("transpose: row groups need a power of two of at least 2^19 rows");
  if (w == 0) return;
  hipEvent_t ev = ctx.prof_begin(K_TRANSPOSE);
  const size_t bx = h >> 11, col_tiles = (w + 15) / 16;  // an eighth of the h / 256 blocks
  size_t by = bx < 1024 ? std::min(col_tiles, (1024 + bx - 1) / bx) : 1;
  const size_t tiles_per_block = (col_tiles + by - 1) / by;
  by = (col_tiles + tiles_per_block - 1) / tiles_per_block;
  hipLaunchKernelGGL(transpose_in_br_k, dim3((unsigned)bx, (unsigned)by), dim3(256), 0, ctx.stream, rowmajor, colmajor, h, w, logh,
                     (u32)(tiles_per_block * 16), 3u, 8u, bitrev32(part, 3));
  HIP_CHECK(hipGetLastError());
  ctx.prof_end(K_TRANSPOSE, ev, 2.0 * double(h) * double(w));
}

namespace {
// four values per thread: one 4 / 8 / 16-byte load, two 16-byte stores
template <class T>
struct Vec4;
template <>
struct Vec4<uint8_t> {
  typedef uchar4 type;
};
template <>
struct Vec4<uint16_t> {
  typedef ushort4 type;
};
template <>
struct Vec4<uint32_t> {
  typedef uint4 type;
};
template <class T>
__global__ __launch_bounds__(256) void widen_k(const T* __restrict__ in, size_t count, u64* __restrict__ out) {
  const size_t i = (blockIdx.x * size_t(blockDim.x) + threadIdx.x) * 4;
  if (i + 4 <= count) {
    const typename Vec4<T>::type q = reinterpret_cast<const typename Vec4<T>::type*>(in)[i >> 2];
    ulonglong2 a, b;
    a.x = q.x, a.y = q.y, b.x = q.z, b.y = q.w;
    *reinterpret_cast<ulonglong2*>(out + i) = a;
    *reinterpret_cast<ulonglong2*>(out + i + 2) = b;
  } else {
    for (size_t k = i; k < count; k++) out[k] = in[k];
  }
}
// The same, reading PINNED HOST memory itself (zero-copy over PCIe): one launch per chunk instead of a DMA copy into a staging
// buffer plus a widening launch behind it. 16-byte loads, a fixed grid that keeps about a megabyte of requests in flight;
// tools/micro/pull_rate.hip: eight 1.8 MB chunks in 327 us against 465 us for copy + widen (one 14.7 MB chunk: 282 against 321).
// (RUNS: the source is a sequence of runs of run_words words - a multiple of 16 - and run m goes to out + m * run_stride)
template <class T, bool RUNS>
__global__ __launch_bounds__(256) void pull_widen_k(const T* __restrict__ host, size_t count, u64* __restrict__ out, u32 run_words, u32 run_stride) {
  constexpr size_t PER = 16 / sizeof(T);
  const size_t stride = size_t(gridDim.x) * blockDim.x * PER;
  const size_t whole = count / PER * PER;
  for (size_t i = (blockIdx.x * size_t(blockDim.x) + threadIdx.x) * PER; i < whole; i += stride) {
    const uint4 v = *reinterpret_cast<const uint4*>(host + i);
    const u32 w[4] = {v.x, v.y, v.z, v.w};
    const size_t dst = RUNS ? size_t((u32)i / run_words) * run_stride + (u32)i % run_words : i;
    ulonglong2* o = reinterpret_cast<ulonglong2*>(out + dst);
    if (sizeof(T) == 1) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        o[2 * k] = make_ulonglong2(w[k] & 0xff, (w[k] >> 8) & 0xff);
        o[2 * k + 1] = make_ulonglong2((w[k] >> 16) & 0xff, w[k] >> 24);
      }
    } else if (sizeof(T) == 2) {
#pragma unroll
      for (int k = 0; k < 4; k++) o[k] = make_ulonglong2(w[k] & 0xffff, w[k] >> 16);
    } else {
      o[0] = make_ulonglong2(w[0], w[1]);
      o[1] = make_ulonglong2(w[2], w[3]);
    }
  }
  if (!RUNS && blockIdx.x == 0 && threadIdx.x == 0)  // (runs are whole multiples of 16 words: no tail)
    for (size_t k = whole; k < count; k++) out[k] = host[k];
}
template <bool RUNS>
void pull_widen_launch(const uint8_t* host_packed, unsigned bytes, size_t count, u64* out, u32 run_words, u32 run_stride, hipStream_t stream) {
  if (reinterpret_cast<uintptr_t>(host_packed) & 15) throw std::runtime_error("pull_widen_words: source must be 16-byte aligned");
  const dim3 grid((unsigned)std::min<size_t>(256, (count * bytes / 16 + 255) / 256 + 1)), block(256);
  if (bytes == 1)
    hipLaunchKernelGGL((pull_widen_k<uint8_t, RUNS>), grid, block, 0, stream, host_packed, count, out, run_words, run_stride);
  else if (bytes == 2)
    hipLaunchKernelGGL((pull_widen_k<uint16_t, RUNS>), grid, block, 0, stream, reinterpret_cast<const uint16_t*>(host_packed), count, out, run_words, run_stride);
  else if (bytes == 4)
    hipLaunchKernelGGL((pull_widen_k<uint32_t, RUNS>), grid, block, 0, stream, reinterpret_cast<const uint32_t*>(host_packed), count, out, run_words, run_stride);
  else
    throw std::runtime_error("pull_widen_words: unsupported width");
  HIP_CHECK(hipGetLastError());
}
}  // namespace
// `host_packed`: pinned host memory (hipHostMalloc), 16-byte aligned
void pull_widen_words(const uint8_t* host_packed, unsigned bytes, size_t count, u64* out, hipStream_t stream) {
  if (!count) return;
  pull_widen_launch<false>(host_packed, bytes, count, out, 0, 0, stream);
}
// the same for a source made of runs of `run_words` words (a multiple of 16; count a multiple of it and below 2^32): run m is
// written to out + m * run_stride
void pull_widen_runs(const uint8_t* host_packed, unsigned bytes, size_t count, u64* out, size_t run_words, size_t run_stride, hipStream_t stream) {
  if (!count) return;
  if (run_words == 0 || run_words % 16 || count % run_words || count >> 32 || run_stride >> 32 || run_stride % 2)
    throw std::runtime_error("pull_widen_runs: runs must be multiples of 16 words, the chunk below 2^32 words");
  pull_widen_launch<true>(host_packed, bytes, count, out, (u32)run_words, (u32)run_stride, stream);
}
void widen_words(const uint8_t* packed, unsigned bytes, size_t count, u64* out, hipStream_t stream) {
  if (!count) return;
  const dim3 grid((unsigned)((count + 1023) / 1024)), block(256);
  if (bytes == 1)
    hipLaunchKernelGGL(widen_k<uint8_t>, grid, block, 0, stream, packed, count, out);
  else if (bytes == 2)
    hipLaunchKernelGGL(widen_k<uint16_t>, grid, block, 0, stream, reinterpret_cast<const uint16_t*>(packed), count, out);
  else if (bytes == 4)
    hipLaunchKernelGGL(widen_k<uint32_t>, grid, block, 0, stream, reinterpret_cast<const uint32_t*>(packed), count, out);
  else
    throw std::runtime_error("widen_words: unsupported width");
  HIP_CHECK(hipGetLastError());
}

void transpose_out(Ctx& ctx, const u64* colmajor, u64* rowmajor, size_t h, size_t w, bool bitrev_rows) {
  if (h * w == 0) return;
  hipLaunchKernelGGL(transpose_out_k, dim3((unsigned)((h * w + 255) / 256)), dim3(256), 0, ctx.stream, colmajor, rowmajor, h,
                     w, log2_strict(h), bitrev_rows ? 1 : 0);
  HIP_CHECK(hipGetLastError());
}

}  // namespace msamd
