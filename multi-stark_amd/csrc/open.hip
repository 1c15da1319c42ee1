// PCS opening on device: inverse denominators, barycentric evaluation, DEEP reduced openings, FRI folding,
// FRI leaf hashing and the query-phase gathers.
// Replaces the device-side work of p3 TwoAdicFriPcs::open as called from /root/reference/src/prover.rs:580
// (rounds built at :540-579): interpolate_coset, the reduced-opening pass, prover::commit_phase fold_matrix,
// and mmcs.open_batch for the queries. The transcript itself stays on the host (prover.hip).
#include "b3_dev.h"
#include "b3_quad.h"
#include "challenge_dev.h"
#include "fri_dev.h"
#include "outer_dev.h"
#include "tree_dev.h"
#include "msamd.h"

namespace msamd {

namespace {

__device__ __forceinline__ u64 coset_point(const u64* __restrict__ t0, const u64* __restrict__ t1, u32 i, unsigned log_h) {
  // x_i = 7 * w_H^{bitrev(i)}
  u32 e = bitrev32(i, log_h) << (TW_LOG - log_h);
  return gl_mul_small(gl_mul(t1[e >> TW_HALF], t0[e & ((1u << TW_HALF) - 1)]), 7);
}

constexpr int DEN_CHUNK = 32;
constexpr int DEN_LOG_CHUNK = 5;
// out[i] = 1 / (z - x_i) for i < H; xout[i] = x_i / (z - x_i) for i < n_x (the barycentric weights: only the
// first H / blowup storage rows, i.e. the trace-domain coset, are ever used there)
// (zp != nullptr: the point is read from device memory - the device transcript sampled it, outer.hip)
//
// 1 / (z - x) = conj(z - x) / N, N = (z0 - x)^2 - 7 z1^2 in the base field, and a thread inverts the norms of its 32
// elements together (Montgomery's trick: one base-field inversion, 72 multiplications, per 32 elements). Element k of a
// thread is base + 256 k, so every store is coalesced; its point is x_base times a constant that depends on k alone
// (bit-reversed storage: the bits of k land in a fixed field of the exponent), read through the scalar unit. Both passes
// are ROLLED loops with the prefix products in LDS (the norms are recomputed on the way back): fully unrolled, the kernel
// was 80 KB of straight-line code at 274 registers - one wave per SIMD, waiting on the instruction cache.
// (row0, H: the launch covers storage rows [row0, H) of the 2^log_h-point domain - a rank of the joint prover needs the
// denominators of its own row range only)
__global__ __launch_bounds__(256) void inv_denoms_k(E2 zv, const E2* __restrict__ zp, unsigned log_h, const u64* __restrict__ t0,
                                                    const u64* __restrict__ t1, E2* __restrict__ out, E2* __restrict__ xout, size_t n_x,
                                                    size_t row0, size_t H) {
  __shared__ u64 pre[DEN_CHUNK][256];
  const E2 z = zp ? *zp : zv;
  const u32 tid = threadIdx.x;
  const size_t base = row0 + blockIdx.x * size_t(256 * DEN_CHUNK) + tid;
  if (base >= H) return;
  const u64 nb = gl_mul_small(gl_sqr(z.c1), (u32)GL_EXT_W);  // 7 z1^2
  const u64 nz1 = gl_neg(z.c1);
  const u64 x0 = coset_point(t0, t1, (u32)base, log_h);
  // bitrev(base + 256 k) = bitrev(base) + bitrev(256 k) when H >= 256 * 32 (disjoint bit fields), and
  // w_H^bitrev(256 k) = w_(2^13)^rev5(k) is a single entry of the upper table; shorter domains take the direct route
  static_assert(TW_LOG - 8 - DEN_LOG_CHUNK >= TW_HALF, "the constant must be a single table entry");
  const bool fast = log_h >= 8 + DEN_LOG_CHUNK && (row0 & (size_t(256 * DEN_CHUNK) - 1)) == 0;
  auto point = [&](int k) -> u64 {
    if (k == 0) return x0;
    if (fast) return gl_mul(x0, t1[(__brev((u32)k) >> (32 - DEN_LOG_CHUNK)) << (TW_LOG - 8 - DEN_LOG_CHUNK - TW_HALF)]);
    return coset_point(t0, t1, (u32)(base + size_t(k) * 256), log_h);
  };
  const size_t left = (H - base + 255) / 256;
  const int cnt = left < (size_t)DEN_CHUNK ? (int)left : DEN_CHUNK;
  u64 acc = 1;
#pragma unroll 1
  for (int k = 0; k < cnt; k++) {
    const u64 d0 = gl_sub(z.c0, point(k));
    pre[k][tid] = acc;
    acc = gl_mul(acc, gl_sub(gl_sqr(d0), nb));
  }
  u64 inv = gl_inv(acc);
#pragma unroll 1
  for (int k = cnt - 1; k >= 0; k--) {
    const size_t i = base + size_t(k) * 256;
    const u64 x = point(k);
    const u64 d0 = gl_sub(z.c0, x);
    const u64 ni = gl_mul(inv, pre[k][tid]);
    inv = gl_mul(inv, gl_sub(gl_sqr(d0), nb));
    const E2 d = e2(gl_mul(d0, ni), gl_mul(nz1, ni));
    out[i] = d;
    if (i < n_x) xout[i] = e2_mul_base(d, x);
  }
}

// Storage index of the point that lies `dec` steps of the domain's generator BEFORE storage index j (bit-reversed storage of a
// coset of 2^log_h points): x_{sigma(j)} = x_j * w^-dec. For zeta' = zeta * g with g = w^dec (g the generator of the trace
// domain, dec = the blowup) that gives 1 / (zeta' - x_j) = g^-1 / (zeta - x_sigma(j)) and
// x_j / (zeta' - x_j) = x_sigma(j) / (zeta - x_sigma(j)): the second opening point of a matrix needs no denominators of
// its own, it reads the first point's through sigma. Consecutive j map to consecutive sigma(j) except where the borrow runs
// through the whole index, so the accesses stay coalesced.
__device__ __forceinline__ size_t rev_dec(size_t j, unsigned log_h, u32 dec) {
  if (log_h == 0) return j;
  const u32 n = __brev((u32)j) >> (32 - log_h);
  const u32 m = (n - dec) & (u32)((size_t(1) << log_h) - 1);
  return __brev(m) >> (32 - log_h);
}

__device__ __forceinline__ u64 wave_sum(u64 v) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    u64 o = (u64)__shfl_xor((unsigned long long)v, m, 64);
    v = gl_add(v, o);
  }
  return v;
}

constexpr int BARY_T = 128;     // threads per workgroup
constexpr int BARY_ROWS = 32;  // rows per thread (the reduction of a thread's accumulators costs as much as ~6 rows of products)
constexpr int BARY_COLS = 2;   // columns per workgroup: their split accumulators (gl_dev.h GlAccS) stay in registers for the whole block
// (round 3: 2, not 4 - with four the kernel needed 266 registers, ONE wave per SIMD, and sat at 0.37 of the issue rate: 149 -> 115 us
// per proof; one column re-reads the weights too often: 137; 16 or 64 rows per thread: the same 115)
// partial[(blk * w + c) * np + p] = sum over the block's rows of col_c[i] * xden_p[i], xden_p[i] = x_i / (z_p - x_i).
// grid = (row blocks, column groups): every thread walks BARY_ROWS rows of BARY_COLS columns, two rows per step with
// all of the step's loads issued before its products (at 2-3 waves per SIMD the loop is otherwise a chain of load
// latencies), accumulates unreduced, and the cross-lane reduction happens once per block through LDS.
template <int NP>
__device__ __forceinline__ void bary_partial_body(u64 (*sh)[BARY_T + 1], const u32 lin, const u64* __restrict__ mat, size_t mat_h, u32 w,
                                                  unsigned log_h, const E2* __restrict__ xden0, const E2* __restrict__ xden1,
                                                  E2* __restrict__ partial, u32 nblk, u32 ngrp, u32 next1) {
  constexpr int NV = BARY_COLS * NP * 2;  // sums per thread
  const size_t h = size_t(1) << log_h;
  // Every column group of a row block reads the same weights (32 bytes per row against 8 per column): the linear
  // workgroup id is decoded so that the groups of one row block follow each other on ONE XCD (workgroups are dealt
  // round-robin to the 8 XCDs) and the weights enter that XCD's L2 once.
  const u32 blk = (lin / (8 * ngrp)) * 8 + (lin & 7);
  if (blk >= nblk) return;
  const u32 grp = (lin >> 3) % ngrp;
  const size_t base = blk * size_t(BARY_T * BARY_ROWS);
  const u32 c0 = grp * BARY_COLS;
  const u32 nc = w - c0 < (u32)BARY_COLS ? w - c0 : (u32)BARY_COLS;
  const u64* __restrict__ col0 = mat + size_t(c0) * mat_h;
  GlAccS acc[BARY_COLS][NP * 2];
#pragma unroll
  for (int c = 0; c < BARY_COLS; c++)
#pragma unroll
    for (int j = 0; j < NP * 2; j++) accs_init(acc[c][j]);
  // software pipeline: the loads of step k + 1 are issued before the products of step k
  E2 xd[2][2][NP];
  u64 v[2][2][BARY_COLS];
  // every load is unconditional (clamped addresses, data selected afterwards): a load under a branch would make the
  // compiler wait for ALL outstanding loads before the products, and the prefetch would hide nothing. Columns past the
  // matrix's width re-read column 0; their sums are never written.
  auto fetch = [&](int k, int buf) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const size_t i = base + size_t(k + u) * BARY_T + threadIdx.x;
      const bool in = i < h;  // rows past the end contribute zeros
      const size_t ii = in ? i : 0;
      const E2 a = xden0[ii];
      xd[buf][u][0] = e2(in ? a.c0 : 0, in ? a.c1 : 0);
      if (NP == 2) {
        const E2 b = xden1[next1 ? rev_dec(ii, log_h, 1) : ii];  // next1: the second point is the first times the trace generator
        xd[buf][u][NP - 1] = e2(in ? b.c0 : 0, in ? b.c1 : 0);
      }
#pragma unroll
      for (int c = 0; c < BARY_COLS; c++) v[buf][u][c] = col0[size_t((u32)c < nc ? c : 0) * mat_h + ii];
    }
  };
  fetch(0, 0);
#pragma unroll
  for (int k = 0; k < BARY_ROWS; k += 2) {
    const int buf = (k >> 1) & 1;
    if (k + 2 < BARY_ROWS) fetch(k + 2, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);  // the scheduler would otherwise sink the prefetch down to its first use
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int c = 0; c < BARY_COLS; c++)
#pragma unroll
        for (int p = 0; p < NP; p++) {
          accs_mad(acc[c][2 * p], xd[buf][u][p].c0, v[buf][u][c]);
          accs_mad(acc[c][2 * p + 1], xd[buf][u][p].c1, v[buf][u][c]);
        }
  }
#pragma unroll
  for (int c = 0; c < BARY_COLS; c++)
#pragma unroll
    for (int j = 0; j < NP * 2; j++) sh[c * NP * 2 + j][threadIdx.x] = accs_reduce(acc[c][j]);
  __syncthreads();
  // BARY_T / NV threads per sum, NV entries each, then a shuffle tree inside the group
  constexpr int TPV = BARY_T / NV;
  const u32 val = threadIdx.x / TPV, part = threadIdx.x % TPV;
  u64 s = 0;
#pragma unroll
  for (int e = 0; e < NV; e++) s = gl_add(s, sh[val][e * TPV + part]);
#pragma unroll
  for (int m = TPV / 2; m > 0; m >>= 1) s = gl_add(s, (u64)__shfl_xor((unsigned long long)s, m, 64));
  const u32 c = val / (NP * 2), j = val % (NP * 2);
  if (part == 0 && c < nc) {
    u64* dst = reinterpret_cast<u64*>(partial + (size_t(blk) * w + c0 + c) * NP);
    dst[j] = s;
  }
}
template <int NP>
__global__ __launch_bounds__(BARY_T) void bary_partial_k(const u64* __restrict__ mat, size_t mat_h, u32 w, unsigned log_h,
                                                      const E2* __restrict__ xden0, const E2* __restrict__ xden1,
                                                      E2* __restrict__ partial, u32 nblk, u32 ngrp, u32 next1) {
  __shared__ u64 sh[BARY_COLS * NP * 2][BARY_T + 1];
  bary_partial_body<NP>(sh, blockIdx.x, mat, mat_h, w, log_h, xden0, xden1, partial, nblk, ngrp, next1);
}

// Several matrices in ONE launch. A matrix's grid is a single round of workgroups at two or three waves per SIMD - its
// duration is the latency of one workgroup's loop, whatever the matrix - so the matrices of an opening (the three
// commitments, the preprocessed trace) run beside each other instead of one after the other.
constexpr int BARY_MAX_JOBS = 8;
struct BaryJob {
  const u64* mat;
  size_t mat_h;
  const E2 *xden0, *xden1;
  E2* partial;   // nblk x w x np
  E2* out;       // w x np
  u32 w, log_h, nblk, ngrp, next1, np;
  u32 wg_begin;  // first workgroup of the partial launch
  u32 out_begin; // first output of the final launch
};
struct BaryBatch {
  BaryJob job[BARY_MAX_JOBS];
  u32 n;
};
__global__ __launch_bounds__(BARY_T) void bary_partial_batch_k(BaryBatch b) {
  __shared__ u64 sh[BARY_COLS * 2 * 2][BARY_T + 1];
  u32 j = 0;
  while (j + 1 < b.n && blockIdx.x >= b.job[j + 1].wg_begin) j++;
  const BaryJob& q = b.job[j];
  const u32 lin = blockIdx.x - q.wg_begin;
  if (q.np == 2)
    bary_partial_body<2>(sh, lin, q.mat, q.mat_h, q.w, q.log_h, q.xden0, q.xden1, q.partial, q.nblk, q.ngrp, q.next1);
  else
    bary_partial_body<1>(sh, lin, q.mat, q.mat_h, q.w, q.log_h, q.xden0, q.xden0, q.partial, q.nblk, q.ngrp, 0u);
}
__global__ __launch_bounds__(64) void bary_final_batch_k(BaryBatch b) {
  u32 j = 0;
  while (j + 1 < b.n && blockIdx.x >= b.job[j + 1].out_begin) j++;
  const BaryJob& q = b.job[j];
  const size_t id = blockIdx.x - q.out_begin;  // over c * np + p
  const size_t stride = size_t(q.w) * q.np;
  u64 s0 = 0, s1 = 0;
  for (size_t k = threadIdx.x; k < q.nblk; k += 64) {
    const E2 v = q.partial[k * stride + id];
    s0 = gl_add(s0, v.c0);
    s1 = gl_add(s1, v.c1);
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if (threadIdx.x == 0) q.out[id] = e2(s0, s1);
}

// one wave per output (c, p): strided sum over the blocks' partials, then a wave reduction
__global__ __launch_bounds__(64) void bary_final_k(const E2* __restrict__ partial, size_t nblk, u32 w, int np, E2* __restrict__ out) {
  const size_t id = blockIdx.x;  // over c * np + p
  u64 s0 = 0, s1 = 0;
  for (size_t b = threadIdx.x; b < nblk; b += 64) {
    E2 v = partial[b * w * np + id];
    s0 = gl_add(s0, v.c0);
    s1 = gl_add(s1, v.c1);
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if (threadIdx.x == 0) out[id] = e2(s0, s1);
}

struct DeepParams {
  const DeepMat* mats;
  u32 nmats;
  const E2* apow;
  DeepPoints pts;
  E2* ro;
  size_t height;
  Digest* leaves;  // when set: digest of FRI row i / 2 = (ro[i], ro[i + 1]), the leaf layer of the first commit-phase round
  u32 log_height;  // log2 of the FULL domain the rows belong to (only read for shifted points)
  size_t row0;     // the launch's row i is row row0 + i of that domain: the inverse denominators are indexed by the full domain's row
  const E2* K_dev; // when set: the points' constants K lie here (filled by open_alpha_k), not in pts.K
};
// ro[i] = sum_q den_q[i] * (K_q - sum_m coeff_{m,q} * s_m[i]),  s_m[i] = sum_c alpha^c m[i][c]
// (= sum over matrices and points of coeff * (red_z - s_m[i]) / (z_q - x_i), regrouped by point so that the
// extension-field products with the inverse denominators happen once per point, not once per matrix and point;
// the products with the constant coeff go through the lazy accumulators like the column sums)
__global__ __launch_bounds__(256) void deep_reduce_k(DeepParams p) {
  // two consecutive rows per thread: every column access is one 16-byte load
  const size_t i = (blockIdx.x * size_t(blockDim.x) + threadIdx.x) * 2;
  if (i >= p.height) return;
  GlAcc T[2][4];  // [point][row * 2 + coordinate]
#pragma unroll
  for (int q = 0; q < 2; q++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc_init(T[q][j]);
  const E2* __restrict__ apow = p.apow;
  for (u32 m = 0; m < p.nmats; m++) {
    const DeepMat& dm = p.mats[m];
    // sum_c alpha^c * m[i][c]: base x ext terms, accumulated unreduced (one reduction per coordinate)
    GlAccS a00, a01, a10, a11;
    accs_init(a00);
    accs_init(a01);
    accs_init(a10);
    accs_init(a11);
    // (the matrix pointer comes out of a descriptor in memory: tell the compiler it is a device allocation, or it emits
    // flat loads, which also count against the LDS counter)
    typedef unsigned long long Pair __attribute__((ext_vector_type(2)));
    typedef const Pair __attribute__((address_space(1))) * GlobalPair;
    const u64* __restrict__ md = dm.d + i;
    const u32 mw = dm.w;
    const size_t mstride = dm.stride ? (size_t)dm.stride : p.height;
    u32 c = 0;
    for (; c + 8 <= mw; c += 8) {  // eight 16-byte loads in flight per lane before the first use
      Pair v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = *(GlobalPair)(md + size_t(c + u) * mstride);
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const E2 a = apow[c + u];
        accs_mad(a00, a.c0, v[u].x);
        accs_mad(a01, a.c1, v[u].x);
        accs_mad(a10, a.c0, v[u].y);
        accs_mad(a11, a.c1, v[u].y);
      }
    }
    for (; c < mw; c++) {
      const Pair v = *(GlobalPair)(md + size_t(c) * mstride);
      const E2 a = apow[c];
      accs_mad(a00, a.c0, v.x);
      accs_mad(a01, a.c1, v.x);
      accs_mad(a10, a.c0, v.y);
      accs_mad(a11, a.c1, v.y);
    }
    const u64 s00 = accs_reduce(a00), s01 = accs_reduce(a01), s10 = accs_reduce(a10), s11 = accs_reduce(a11);
    for (u32 k = 0; k < dm.npoints; k++) {
      // (c0 + c1 X)(s0 + s1 X) = c0 s0 + 7 c1 s1 + (c0 s1 + c1 s0) X
      const u64 c0 = dm.coeff[k].c0, c1 = dm.coeff[k].c1, c7 = dm.coeff7[k];
#pragma unroll
      for (int q = 0; q < 2; q++) {
        if (dm.pt[k] == (u32)q) {  // uniform across the launch
          acc_mad(T[q][0], c0, s00);
          acc_mad(T[q][0], c7, s01);
          acc_mad(T[q][1], c0, s01);
          acc_mad(T[q][1], c1, s00);
          acc_mad(T[q][2], c0, s10);
          acc_mad(T[q][2], c7, s11);
          acc_mad(T[q][3], c0, s11);
          acc_mad(T[q][3], c1, s10);
        }
      }
    }
  }
  E2 r0 = e2(0), r1 = e2(0);
#pragma unroll
  for (int q = 0; q < 2; q++) {
    if ((u32)q < p.pts.n) {
      const E2 K = p.K_dev ? p.K_dev[q] : p.pts.K[q];
      const E2 t0 = e2(gl_sub(K.c0, acc_reduce(T[q][0])), gl_sub(K.c1, acc_reduce(T[q][1])));
      const E2 t1 = e2(gl_sub(K.c0, acc_reduce(T[q][2])), gl_sub(K.c1, acc_reduce(T[q][3])));
      const E2* __restrict__ den = p.pts.den[q];
      const u32 dec = p.pts.shift[q];  // 0, or the point is an earlier one times w^dec: read that one's denominators (rev_dec)
      const size_t g = p.row0 + i;  // row of the full domain (a joint proof reduces a row range: prover_sharded.inc)
      r0 = e2_add(r0, e2_mul(t0, den[dec ? rev_dec(g, p.log_height, dec) : g]));
      r1 = e2_add(r1, e2_mul(t1, den[dec ? rev_dec(g + 1, p.log_height, dec) : g + 1]));
    }
  }
  p.ro[i] = r0;
  p.ro[i + 1] = r1;
  if (p.leaves) {
    // the pair this thread has just produced is one row of FRI's first committed matrix: hash it here (the pass waits on
    // memory, the 680 instructions ride along) instead of reading the vector back in a launch of its own
    u32 m[16];
    m[0] = (u32)r0.c0;
    m[1] = (u32)(r0.c0 >> 32);
    m[2] = (u32)r0.c1;
    m[3] = (u32)(r0.c1 >> 32);
    m[4] = (u32)r1.c0;
    m[5] = (u32)(r1.c0 >> 32);
    m[6] = (u32)r1.c1;
    m[7] = (u32)(r1.c1 >> 32);
#pragma unroll
    for (int k = 8; k < 16; k++) m[k] = 0;
    u32 cv[8];
    b3_iv(cv);
    b3_compress(cv, m, 0, 32, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
    uint4* q = reinterpret_cast<uint4*>(p.leaves + (i >> 1));
    q[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    q[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
  }
}

// The same for FEW rows of VERY wide matrices (the reference's BLAKE3 system: a 1024-row LDE of 2625 + 146 + 2 columns): with a
// thread per row pair a few hundred threads walk thousands of columns each, one dependent batch of loads after the other.
// Here 16 row pairs x 16 column slices form a workgroup: slice s takes the columns c = s (mod 16) of every matrix (a fixed
// column is still one 256-byte run over the row pairs), the per-point sums are LINEAR in the column sums, so every slice
// carries partial sums through to the reduced per-point values, LDS adds the 16 partials, and slice 0 finishes the row pair
// exactly as deep_reduce_k does (same products, same order of the final additions: bit-identical).
__global__ __launch_bounds__(256) void deep_reduce_wide_k(DeepParams p) {
  __shared__ u64 part[16][8][16];  // [slice][point * 4 + row * 2 + coordinate][row pair]
  const u32 rp = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const size_t i = (blockIdx.x * size_t(16) + rp) * 2;
  const bool live = i < p.height;
  GlAcc T[2][4];
#pragma unroll
  for (int q = 0; q < 2; q++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc_init(T[q][j]);
  const E2* __restrict__ apow = p.apow;
  if (live) {
    for (u32 m = 0; m < p.nmats; m++) {
      const DeepMat& dm = p.mats[m];
      GlAccS a00, a01, a10, a11;
      accs_init(a00);
      accs_init(a01);
      accs_init(a10);
      accs_init(a11);
      typedef unsigned long long Pair __attribute__((ext_vector_type(2)));
      typedef const Pair __attribute__((address_space(1))) * GlobalPair;
      const u64* __restrict__ md = dm.d + i;
      const u32 mw = dm.w;
      const size_t mstride = dm.stride ? (size_t)dm.stride : p.height;
      u32 c = slice;
      for (; c + 16 * 3 < mw; c += 16 * 4) {  // four 16-byte loads in flight per lane
        Pair v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = *(GlobalPair)(md + size_t(c + 16 * u) * mstride);
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const E2 a = apow[c + 16 * u];
          accs_mad(a00, a.c0, v[u].x);
          accs_mad(a01, a.c1, v[u].x);
          accs_mad(a10, a.c0, v[u].y);
          accs_mad(a11, a.c1, v[u].y);
        }
      }
      for (; c < mw; c += 16) {
        const Pair v = *(GlobalPair)(md + size_t(c) * mstride);
        const E2 a = apow[c];
        accs_mad(a00, a.c0, v.x);
        accs_mad(a01, a.c1, v.x);
        accs_mad(a10, a.c0, v.y);
        accs_mad(a11, a.c1, v.y);
      }
      const u64 s00 = accs_reduce(a00), s01 = accs_reduce(a01), s10 = accs_reduce(a10), s11 = accs_reduce(a11);
      for (u32 k = 0; k < dm.npoints; k++) {
        const u64 c0 = dm.coeff[k].c0, c1 = dm.coeff[k].c1, c7 = dm.coeff7[k];
#pragma unroll
        for (int q = 0; q < 2; q++) {
          if (dm.pt[k] == (u32)q) {
            acc_mad(T[q][0], c0, s00);
            acc_mad(T[q][0], c7, s01);
            acc_mad(T[q][1], c0, s01);
            acc_mad(T[q][1], c1, s00);
            acc_mad(T[q][2], c0, s10);
            acc_mad(T[q][2], c7, s11);
            acc_mad(T[q][3], c0, s11);
            acc_mad(T[q][3], c1, s10);
          }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 2; q++)
#pragma unroll
    for (int j = 0; j < 4; j++) part[slice][q * 4 + j][rp] = acc_reduce(T[q][j]);
  __syncthreads();
  if (slice != 0 || !live) return;
  u64 tot[8];
#pragma unroll
  for (int v = 0; v < 8; v++) {
    u64 a = part[0][v][rp];
    for (int sl = 1; sl < 16; sl++) a = gl_add(a, part[sl][v][rp]);
    tot[v] = a;
  }
  E2 r0 = e2(0), r1 = e2(0);
#pragma unroll
  for (int q = 0; q < 2; q++) {
    if ((u32)q < p.pts.n) {
      const E2 K = p.K_dev ? p.K_dev[q] : p.pts.K[q];
      const E2 t0 = e2(gl_sub(K.c0, tot[q * 4 + 0]), gl_sub(K.c1, tot[q * 4 + 1]));
      const E2 t1 = e2(gl_sub(K.c0, tot[q * 4 + 2]), gl_sub(K.c1, tot[q * 4 + 3]));
      const E2* __restrict__ den = p.pts.den[q];
      const u32 dec = p.pts.shift[q];
      const size_t g = p.row0 + i;  // row of the full domain (a joint proof reduces a row range: prover_sharded.inc)
      r0 = e2_add(r0, e2_mul(t0, den[dec ? rev_dec(g, p.log_height, dec) : g]));
      r1 = e2_add(r1, e2_mul(t1, den[dec ? rev_dec(g + 1, p.log_height, dec) : g + 1]));
    }
  }
  p.ro[i] = r0;
  p.ro[i + 1] = r1;
  if (p.leaves) {
    u32 m[16];
    m[0] = (u32)r0.c0;
    m[1] = (u32)(r0.c0 >> 32);
    m[2] = (u32)r0.c1;
    m[3] = (u32)(r0.c1 >> 32);
    m[4] = (u32)r1.c0;
    m[5] = (u32)(r1.c0 >> 32);
    m[6] = (u32)r1.c1;
    m[7] = (u32)(r1.c1 >> 32);
#pragma unroll
    for (int k = 8; k < 16; k++) m[k] = 0;
    u32 cv[8];
    b3_iv(cv);
    b3_compress(cv, m, 0, 32, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
    uint4* q = reinterpret_cast<uint4*>(p.leaves + (i >> 1));
    q[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    q[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
  }
}

// out[i] = (1/2 + pw) lo + (1/2 - pw) hi, pw = (beta/2) w_{2R}^{-bitrev(i)}; optional roll-in out[i] += f * in[i]
// (row0 != 0: `cur` / `roll` / `out` are a rank's slice of the layer, which starts at row `row0` of 2^log_rows)
__global__ __launch_bounds__(256) void fri_fold_k(const E2* __restrict__ cur, size_t rows, unsigned log_rows, E2 half_beta, u64 half,
                                                  const E2* __restrict__ roll, E2 roll_f, const u64* __restrict__ t0i,
                                                  const u64* __restrict__ t1i, E2* __restrict__ out, size_t row0) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (i >= rows) return;
  u32 e = bitrev32((u32)(i + row0), log_rows) << (TW_LOG - log_rows - 1);
  u64 gp = gl_mul(t1i[e >> TW_HALF], t0i[e & ((1u << TW_HALF) - 1)]);
  E2 pw = e2_mul_base(half_beta, gp);
  (void)half;
  E2 r = fri_fold_value(cur[2 * i], cur[2 * i + 1], pw);
  if (roll) r = e2_add(r, e2_mul(roll_f, roll[i]));
  out[i] = r;
}

// leaf digest of FRI row i = BLAKE3 of the 32 bytes (lo.c0, lo.c1, hi.c0, hi.c1) (ExtensionMmcs flattening)
__global__ __launch_bounds__(256) void fri_leaf_hash_k(const E2* __restrict__ cur, size_t rows, Digest* __restrict__ out) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (i >= rows) return;
  E2 lo = cur[2 * i], hi = cur[2 * i + 1];
  u32 m[16];
  m[0] = (u32)lo.c0;
  m[1] = (u32)(lo.c0 >> 32);
  m[2] = (u32)lo.c1;
  m[3] = (u32)(lo.c1 >> 32);
  m[4] = (u32)hi.c0;
  m[5] = (u32)(hi.c0 >> 32);
  m[6] = (u32)hi.c1;
  m[7] = (u32)(hi.c1 >> 32);
#pragma unroll
  for (int k = 8; k < 16; k++) m[k] = 0;
  u32 cv[8];
  b3_iv(cv);
  b3_compress(cv, m, 0, 32, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
  uint4* q = reinterpret_cast<uint4*>(out + i);
  q[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
  q[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}

// the same for a round of arity 2^log_arity >= 4 (FriParameters::max_log_arity > 1): row i is the 2^log_arity consecutive
// values cur[i 2^a ..], 16 bytes each - whole 64-byte blocks of ONE chunk (a <= 6, checked by the caller)
__global__ __launch_bounds__(256) void fri_leaf_hash_wide_k(const E2* __restrict__ cur, size_t rows, unsigned log_arity, Digest* __restrict__ out) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (i >= rows) return;
  const u32 nb = 1u << (log_arity - 2);
  const uint4* src = reinterpret_cast<const uint4*>(cur + (i << log_arity));
  u32 cv[8];
  b3_iv(cv);
  for (u32 b = 0; b < nb; b++) {
    u32 m[16];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint4 v = src[4 * b + k];
      m[4 * k] = v.x, m[4 * k + 1] = v.y, m[4 * k + 2] = v.z, m[4 * k + 3] = v.w;
    }
    b3_compress(cv, m, 0, 64, (b == 0 ? B3_CHUNK_START : 0u) | (b == nb - 1 ? (B3_CHUNK_END | B3_ROOT) : 0u));
  }
  uint4* q = reinterpret_cast<uint4*>(out + i);
  q[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
  q[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}

// Fold with the challenge still on the device (rec->beta written by the previous round's challenger step) and,
// when LEAF, the next round's leaf digests in the same pass: thread j produces out[2j], out[2j+1] (one FRI row of
// the next layer) from cur[4j .. 4j+3] and hashes that row.
template <bool LEAF>
__global__ __launch_bounds__(256) void fri_fold_dev_k(const E2* __restrict__ cur, size_t rows, unsigned log_rows,
                                                      const FriTailRound* __restrict__ rec, const E2* __restrict__ roll,
                                                      const u64* __restrict__ t0i, const u64* __restrict__ t1i, E2* __restrict__ out,
                                                      Digest* __restrict__ leaves, size_t row0, u32 squarings) {
  // (row0: cur / roll / out are the slice [row0, row0 + rows) of a layer of 2^log_rows rows - a rank's row range of a joint proof)
  const size_t j = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (2 * j >= rows) return;
  const u64 half = 0x7FFFFFFF80000001ULL;  // 1/2 mod p
  E2 beta = rec->beta;
  for (u32 q = 0; q < squarings; q++) beta = e2_sqr(beta);  // (a later step of a wide round)
  const E2 hb = e2_mul_base(beta, half);
  const E2 rf = e2_sqr(beta);
  E2 o[2];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const size_t i = 2 * j + k;
    o[k] = e2(0);
    if (i < rows) {
      u32 e = bitrev32((u32)(row0 + i), log_rows) << (TW_LOG - log_rows - 1);
      u64 gp = gl_mul(t1i[e >> TW_HALF], t0i[e & ((1u << TW_HALF) - 1)]);
      E2 pw = e2_mul_base(hb, gp);
      E2 r = fri_fold_value(cur[2 * i], cur[2 * i + 1], pw);
      if (roll) r = e2_add(r, e2_mul(rf, roll[i]));
      out[i] = r;
      o[k] = r;
    }
  }
  if (LEAF) {
    u32 m[16];
    m[0] = (u32)o[0].c0;
    m[1] = (u32)(o[0].c0 >> 32);
    m[2] = (u32)o[0].c1;
    m[3] = (u32)(o[0].c1 >> 32);
    m[4] = (u32)o[1].c0;
    m[5] = (u32)(o[1].c0 >> 32);
    m[6] = (u32)o[1].c1;
    m[7] = (u32)(o[1].c1 >> 32);
#pragma unroll
    for (int k = 8; k < 16; k++) m[k] = 0;
    u32 cv[8];
    b3_iv(cv);
    b3_compress(cv, m, 0, 32, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
    uint4* q = reinterpret_cast<uint4*>(leaves + j);
    q[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
    q[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
  }
}

// proof-of-work search: smallest w in [w0, w0 + n) such that BLAKE3(prefix || w as 8 LE bytes), read as the
// challenger's sample_bits (u64 from the LAST 8 digest bytes, reversed), has its low `bits` bits clear.
// cv_mid = chaining value after the prefix blocks that cannot contain witness bytes; tail = remaining prefix bytes.
struct GrindParams {
  u32 cv_mid[8];
  u32 tail[32];     // tail bytes as LE words, zero padded (tail_len < 120)
  u32 tail_len;     // bytes of prefix in the tail
  u32 first_block;  // 1 if the tail's first block is the chunk's first block
  u64 w0;
  u64 mask;
};
__global__ __launch_bounds__(256) void grind_k(GrindParams p, unsigned long long* __restrict__ best) {
  const u64 w = p.w0 + blockIdx.x * u64(blockDim.x) + threadIdx.x;
  u32 m[32];
#pragma unroll
  for (int i = 0; i < 32; i++) m[i] = p.tail[i];
  for (int k = 0; k < 8; k++) {
    u32 pos = p.tail_len + k;
    m[pos >> 2] |= (u32)((w >> (8 * k)) & 0xff) << (8 * (pos & 3));
  }
  const u32 total = p.tail_len + 8;
  u32 cv[8];
#pragma unroll
  for (int i = 0; i < 8; i++) cv[i] = p.cv_mid[i];
  if (total <= 64) {
    b3_compress(cv, m, 0, total, (p.first_block ? B3_CHUNK_START : 0) | B3_CHUNK_END | B3_ROOT);
  } else {
    b3_compress(cv, m, 0, 64, p.first_block ? B3_CHUNK_START : 0);
    b3_compress(cv, m + 16, 0, total - 64, B3_CHUNK_END | B3_ROOT);
  }
  // digest bytes 31..24 form the sampled u64 little-endian: byte k = d[31 - k]
  const u32 hi = cv[7], lo = cv[6];
  const u64 v = ((u64)__builtin_bswap32(lo) << 32) | (u64)__builtin_bswap32(hi);
  if ((v & p.mask) == 0) atomicMin(best, (unsigned long long)w);
}

// Same search with the commitment still on the device: transcript = prefix (host bytes, 4-byte multiple) || cap
// digests (device) || w, at most 128 bytes. Saves the host round trip between "read the cap" and "grind".
struct GrindCapParams {
  u32 prefix[32];
  u32 prefix_words;
  u32 cap_words;  // 8 per digest
  u64 w0;
  u64 mask;
};
__global__ __launch_bounds__(256) void grind_cap_k(GrindCapParams p, const u32* __restrict__ cap, unsigned long long* __restrict__ best) {
  const u64 w = p.w0 + blockIdx.x * u64(blockDim.x) + threadIdx.x;
  u32 m[32];
#pragma unroll
  for (int i = 0; i < 32; i++) m[i] = p.prefix[i];
  const u32 q = p.prefix_words + p.cap_words;
  for (u32 i = 0; i < p.cap_words; i++) m[p.prefix_words + i] = cap[i];
  m[q] = (u32)w;
  m[q + 1] = (u32)(w >> 32);
  const u32 total = 4 * q + 8;
  u32 cv[8];
  b3_iv(cv);
  if (total <= 64) {
    b3_compress(cv, m, 0, total, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
  } else {
    b3_compress(cv, m, 0, 64, B3_CHUNK_START);
    b3_compress(cv, m + 16, 0, total - 64, B3_CHUNK_END | B3_ROOT);
  }
  const u64 v = ((u64)__builtin_bswap32(cv[6]) << 32) | (u64)__builtin_bswap32(cv[7]);
  if ((v & p.mask) == 0) atomicMin(best, (unsigned long long)w);
}

// ---------------------------------------------------------------------------------------------------------
// FRI tail: every remaining commit-phase round in one single-workgroup launch (vectors of <= 2048 elements).
// (the tail starts at 2048 values at most: the inputs that can still roll in have 2^1 .. 2^10 values - ten distinct heights;
// round 4's fuzzing found the limit of eight that stood here with a twelve-circuit system)
constexpr size_t FRI_TAIL_MAX_ROLLS = 12;
struct FriTailParams {
  const E2* cur0;
  u32 len0, n_rounds, pow_bits, n_roll;
  u32* state;    // device: challenger input buffer (one 32-byte digest) as little-endian words; updated
  FriTailRoll roll[FRI_TAIL_MAX_ROLLS];
  Digest* tree_out;   // round r: rows_r + rows_r/2 + ... + 1 digests, rounds back to back
  E2* layers_out;     // input vectors of rounds 1.. (round 0's input is cur0), back to back
  E2* final_out;
  FriTailRound* rounds;
  const u64 *t0i, *t1i;
};

__global__ __launch_bounds__(1024) void fri_tail_k(FriTailParams p) {
  __shared__ E2 cur[2048];
  __shared__ __attribute__((aligned(16))) u32 tree[1024 * 8];
  __shared__ ChallengeShared cs;
  const u32 t = threadIdx.x;
  u32 len = p.len0;
  for (u32 i = t; i < len; i += 1024) cur[i] = p.cur0[i];
  if (t < 8) cs.st[t] = p.state[t];
  __syncthreads();
  Digest* tout = p.tree_out;
  E2* lout = p.layers_out;
  u32 roll_i = 0;
  while (roll_i < p.n_roll && p.roll[roll_i].len >= len) roll_i++;  // inputs taller than the tail never roll in here
  const u64 half = 0x7FFFFFFF80000001ULL;  // 1/2 mod p
  for (u32 r = 0; r < p.n_rounds; r++) {
    const u32 rows = len >> 1;
    const unsigned log_rows = 31 - __clz(rows);
    // ---- leaf digests (ExtensionMmcs rows of two Ext2 values = 32 bytes) and tree levels. The whole round is a chain
    // of dependent compressions: one lane per node while a level has 128 nodes or more, one quad per node below
    // (tree_dev.h)
    const u32 quad = t >> 2, qc = t & 3;
    u32* gout = reinterpret_cast<u32*>(tout);
    if (rows >= 128) {
      if (t < rows) {
        u32 m[16], cv[8];
        fri_row_block(cur[2 * t], cur[2 * t + 1], m);
        b3_iv(cv);
        b3_compress(cv, m, 0, 32, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
        lds_store_digest(tree, t, cv);
        store_digest(tout + t, cv);
      }
    } else {
      u32 lo = 0, hi = 0;
      if (quad < rows) {
        b3_quad_row32(reinterpret_cast<const u32*>(&cur[2 * quad]), lo, hi);
        tree[8 * quad + qc] = lo;
        tree[8 * quad + 4 + qc] = hi;
        gout[8 * quad + qc] = lo;
        gout[8 * quad + 4 + qc] = hi;
      }
    }
    __syncthreads();
    gout += 8 * rows;
    for (u32 n = rows >> 1; n >= 1; n >>= 1) {
      if (n <= 16) {  // the last levels on one wave (tree_dev.h)
        tree_levels_first_wave(tree, n, reinterpret_cast<Digest*>(gout));
        gout += 8 * (2 * n - 1);
        break;
      }
      tree_level_plain(tree, n, reinterpret_cast<Digest*>(gout));
      gout += 8 * n;
    }
    tout = reinterpret_cast<Digest*>(gout);
    // ---- challenger: observe the root, grind, sample beta
    challenger_round<1024>(cs, tree, p.pow_bits);
    if (t == 0) {
      FriTailRound& out = p.rounds[r];
      for (int k = 0; k < 8; k++) out.root[k] = tree[k];
      out.witness = cs.wit;
      out.beta = cs.beta;
    }
    // ---- fold (+ roll-in of a reduced opening of matching length)
    const E2 beta = cs.beta;
    const E2 hb = e2_mul_base(beta, half);
    const bool do_roll = roll_i < p.n_roll && p.roll[roll_i].len == rows;
    E2 nv = e2(0);
    if (t < rows) {
      u32 e = bitrev32(t, log_rows) << (TW_LOG - log_rows - 1);
      u64 gp = gl_mul(p.t1i[e >> TW_HALF], p.t0i[e & ((1u << TW_HALF) - 1)]);
      E2 pw = e2_mul_base(hb, gp);
      nv = fri_fold_value(cur[2 * t], cur[2 * t + 1], pw);
      if (do_roll) nv = e2_add(nv, e2_mul(e2_sqr(beta), p.roll[roll_i].p[t]));
    }
    __syncthreads();
    if (t < rows) {
      cur[t] = nv;
      if (r + 1 < p.n_rounds) lout[t] = nv;
    }
    if (do_roll) roll_i++;
    lout += rows;
    len = rows;
    __syncthreads();
  }
  if (t < len) p.final_out[t] = cur[t];
  if (t < 8) p.state[t] = cs.st[t];
}

// Query-phase challenger work in one single-workgroup launch, for a one-coefficient final polynomial: the
// transcript block is state (32 bytes) || final coefficient (16 bytes) [|| witness (8 bytes)]; after the
// proof-of-work check every query index is one sample_bits(log_max_height). out[0] = witness, out[1 + q] = index q.
__global__ __launch_bounds__(1024) void fri_query_challenge_k(const u32* __restrict__ state, const E2* __restrict__ fin, u32 pow_bits,
                                                              u32 n_queries, u32 log_max_height, u64* __restrict__ out) {
  __shared__ u32 dg[8];
  __shared__ unsigned long long best;
  const u32 t = threadIdx.x;
  u32 blk[16];
#pragma unroll
  for (int k = 0; k < 8; k++) blk[k] = state[k];
  const E2 f = fin[0];
  blk[8] = (u32)f.c0;
  blk[9] = (u32)(f.c0 >> 32);
  blk[10] = (u32)f.c1;
  blk[11] = (u32)(f.c1 >> 32);
  blk[12] = blk[13] = blk[14] = blk[15] = 0;
  u64 wit = 0;
  if (pow_bits) {
    if (t == 0) best = ~0ull;
    __syncthreads();
    const u64 mask = (u64(1) << pow_bits) - 1;
    u32 cv[8];
    u64 w;
    for (u64 base = 0;; base += 1024) {
      w = base + t;
      blk[12] = (u32)w;
      blk[13] = (u32)(w >> 32);
      b3_iv(cv);
      b3_compress(cv, blk, 0, 56, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
      const u64 v = (u64)__builtin_bswap32(cv[7]) | ((u64)__builtin_bswap32(cv[6]) << 32);
      if ((v & mask) == 0) atomicMin(&best, (unsigned long long)w);
      __syncthreads();
      const unsigned long long b = best;
      __syncthreads();
      if (b != ~0ull) {
        wit = b;
        break;
      }
    }
    if (w == wit) {
#pragma unroll
      for (int k = 0; k < 8; k++) dg[k] = cv[k];
    }
    __syncthreads();
  }
  int pos;
  if (pow_bits) {
    pos = 24;  // check_witness' sample_bits consumed digest bytes 24..31
  } else {
    if (t == 0) {
      u32 cv[8];
      b3_iv(cv);
      b3_compress(cv, blk, 0, 48, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
      for (int k = 0; k < 8; k++) dg[k] = cv[k];
    }
    __syncthreads();
    pos = 32;
  }
  if (t == 0) out[0] = wit;
  // every query index is 8 bytes popped from the back of the output buffer; an exhausted buffer is refilled by
  // digest <- BLAKE3(digest): a chain of dependent compressions, each run by one quad (half the latency of one lane)
  // (only the first wave goes on: the chain's two hand-overs per link then cost a wave barrier, not a barrier of sixteen waves)
  const u64 imask = (u64(1) << log_max_height) - 1;
  if (t >= 64) return;
  for (u32 q = 0; q < n_queries; q++) {
    if (pos == 0) {
      u32 lo = 0, hi = 0;
      if (t < 4) b3_quad_row32(dg, lo, hi);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (t < 4) {
        dg[t] = lo;
        dg[4 + t] = hi;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      pos = 32;
    }
    pos -= 8;
    if (t == 0) out[1 + q] = be64_at(dg, pos) & imask;
  }
}

__global__ void gather_k(const GatherReq* __restrict__ reqs, size_t n, uint8_t* __restrict__ out) {
  size_t r = blockIdx.x;
  if (r >= n) return;
  const GatherReq q = reqs[r];
  if (q.kind == 0) {
    const u64* m = (const u64*)q.base;
    u64* o = (u64*)(out + q.out_off);
    for (u32 c = threadIdx.x; c < q.count; c += blockDim.x) o[c] = m[size_t(c) * q.stride + q.index];
  } else {
    const u32* d = (const u32*)((const Digest*)q.base + q.index);
    u32* o = (u32*)(out + q.out_off);
    for (u32 c = threadIdx.x; c < 8 * q.count; c += blockDim.x) o[c] = d[c];
  }
}

// owner != ~0: only the queries whose index >> owner_shift equals `owner` are served (a rank of a joint proof holds the rows
// of its own row range only: prover_sharded.inc); the blocks of the other queries are left as they are (zeroed by the caller)
__global__ void gather_queries_k(const GatherSeg* __restrict__ segs, const u64* __restrict__ indices, size_t qbytes,
                                 uint8_t* __restrict__ out, u32 owner_shift, u32 owner) {
  const GatherSeg s = segs[blockIdx.x];
  const u64 index = indices[blockIdx.y];
  if (owner != ~0u && (u32)(index >> owner_shift) != owner) return;
  const u64 e = (index >> s.shift) ^ s.flip;
  uint8_t* o = out + size_t(blockIdx.y) * qbytes + s.out_off;
  if (s.kind == 0) {
    const u64* m = (const u64*)s.base;
    u64* ow = (u64*)o;
    for (u32 c = threadIdx.x; c < s.count; c += blockDim.x) ow[c] = m[size_t(c) * s.stride + e];
  } else if (s.kind == 1) {
    const u32* d = (const u32*)((const Digest*)s.base + e);
    if (threadIdx.x < 8) ((u32*)o)[threadIdx.x] = d[threadIdx.x];
  } else if (s.kind == 3) {
    const u64* m = (const u64*)s.base + e * s.count;
    u64* ow = (u64*)o;
    for (u32 c = threadIdx.x; c < s.count; c += blockDim.x) ow[c] = m[c];
  } else {
    const u32* d = (const u32*)((const E2*)s.base + e);
    if (threadIdx.x < 4) ((u32*)o)[threadIdx.x] = d[threadIdx.x];
  }
}

}  // namespace

static void inv_denoms_launch(Ctx& ctx, E2 z, const E2* z_dev, unsigned log_h, E2* out, E2* xout, size_t n_x, size_t row0 = 0,
                              size_t rows = ~size_t(0)) {
  if (log_h > TW_LOG) throw std::runtime_error("LDE height above 2^28 is not supported");
  size_t H = size_t(1) << log_h;
  if (n_x > H) throw std::runtime_error("inv_denoms: weight vector longer than the domain");
  if (rows == ~size_t(0)) rows = H - std::min(row0, H);
  if (row0 > H || rows > H - row0) throw std::runtime_error("inv_denoms: row range outside the domain");
  if (rows == 0) return;
  size_t blocks = (rows + 256 * DEN_CHUNK - 1) / (256 * DEN_CHUNK);
  hipLaunchKernelGGL(inv_denoms_k, dim3((unsigned)blocks), dim3(256), 0, ctx.stream, z, z_dev, log_h, ctx.tw0, ctx.tw1, out, xout,
                     xout ? n_x : size_t(0), row0, row0 + rows);
  HIP_CHECK(hipGetLastError());
}
void inv_denoms_rows(Ctx& ctx, E2 z, unsigned log_h, E2* out, E2* xout, size_t n_x, size_t row0, size_t rows) {
  inv_denoms_launch(ctx, z, nullptr, log_h, out, xout, n_x, row0, rows);
}
void inv_denoms_rows_dev(Ctx& ctx, const E2* z_dev, unsigned log_h, E2* out, E2* xout, size_t n_x, size_t row0, size_t rows) {
  inv_denoms_launch(ctx, e2(0), z_dev, log_h, out, xout, n_x, row0, rows);
}
void inv_denoms(Ctx& ctx, E2 z, unsigned log_h, E2* out, E2* xout, size_t n_x) { inv_denoms_launch(ctx, z, nullptr, log_h, out, xout, n_x); }
void inv_denoms_dev(Ctx& ctx, const E2* z_dev, unsigned log_h, E2* out, E2* xout, size_t n_x) {
  inv_denoms_launch(ctx, e2(0), z_dev, log_h, out, xout, n_x);
}

// launches only: raw sums sum_{i<h} col_c[i] x_i invden_p[i] into `out_dev` (w * npoints values, index c * np + p)
void bary_sums_async(Ctx& ctx, const u64* mat, size_t mat_h, size_t w, unsigned log_h, const E2* xden0, const E2* xden1, int npoints,
                     E2* out_dev, bool second_is_next) {
  if (npoints == 0) return;
  size_t h = size_t(1) << log_h;
  size_t nblk = (h + BARY_T * BARY_ROWS - 1) / (BARY_T * BARY_ROWS);
  size_t ngrp = (w + BARY_COLS - 1) / BARY_COLS;
  const size_t nlin = ((nblk + 7) / 8) * 8 * ngrp;
  if (nlin > 0x7fffffffu) throw std::runtime_error("bary: matrix too large");
  DBuf<E2> partial(ctx, nblk * w * npoints);  // stream-ordered reuse keeps it valid until bary_final_k has run
  hipEvent_t ev = ctx.prof_begin(K_BARY);
  dim3 grid((unsigned)nlin);
  if (npoints == 1)
    hipLaunchKernelGGL(bary_partial_k<1>, grid, dim3(BARY_T), 0, ctx.stream, mat, mat_h, (u32)w, log_h, xden0, xden0, partial.p, (u32)nblk, (u32)ngrp, 0u);
  else
    hipLaunchKernelGGL(bary_partial_k<2>, grid, dim3(BARY_T), 0, ctx.stream, mat, mat_h, (u32)w, log_h, xden0, second_is_next ? xden0 : xden1, partial.p,
                       (u32)nblk, (u32)ngrp, second_is_next ? 1u : 0u);
  size_t tot = w * npoints;
  hipLaunchKernelGGL(bary_final_k, dim3((unsigned)tot), dim3(64), 0, ctx.stream, partial.p, nblk, (u32)w, npoints, out_dev);
  ctx.prof_end(K_BARY, ev, double(h) * 8.0 * w);
  HIP_CHECK(hipGetLastError());
}

// the same for several matrices at once (at most BARY_MAX_JOBS per launch pair; `partial_keep` holds the partial sums)
void bary_sums_batch(Ctx& ctx, const std::vector<BarySpec>& specs, DBuf<E2>& partial_keep) {
  size_t total_partial = 0;
  for (auto& sp : specs) {
    if (sp.npoints < 1 || sp.npoints > 2) throw std::runtime_error("bary: one or two points per matrix");
    const size_t h = size_t(1) << sp.log_h;
    total_partial += ((h + BARY_T * BARY_ROWS - 1) / (BARY_T * BARY_ROWS)) * sp.w * sp.npoints;
  }
  partial_keep = DBuf<E2>(ctx, std::max<size_t>(total_partial, 1));
  size_t poff = 0;
  for (size_t first = 0; first < specs.size(); first += BARY_MAX_JOBS) {
    BaryBatch b;
    memset(&b, 0, sizeof(b));
    u64 wg = 0, outs = 0;
    double bytes = 0;
    for (size_t k = first; k < specs.size() && k < first + BARY_MAX_JOBS; k++) {
      const BarySpec& sp = specs[k];
      BaryJob& q = b.job[b.n++];
      const size_t h = size_t(1) << sp.log_h;
      const size_t nblk = (h + BARY_T * BARY_ROWS - 1) / (BARY_T * BARY_ROWS), ngrp = (sp.w + BARY_COLS - 1) / BARY_COLS;
      q.mat = sp.mat;
      q.mat_h = sp.mat_h;
      q.xden0 = sp.xden0;
      q.xden1 = sp.second_is_next ? sp.xden0 : sp.xden1;
      q.partial = partial_keep.p + poff;
      q.out = sp.out_dev;
      q.w = (u32)sp.w;
      q.log_h = sp.log_h;
      q.nblk = (u32)nblk;
      q.ngrp = (u32)ngrp;
      q.next1 = sp.second_is_next ? 1u : 0u;
      q.np = (u32)sp.npoints;
      q.wg_begin = (u32)wg;
      q.out_begin = (u32)outs;
      poff += nblk * sp.w * sp.npoints;
      wg += ((nblk + 7) / 8) * 8 * ngrp;
      outs += sp.w * sp.npoints;
      bytes += double(h) * 8.0 * sp.w;
    }
    if (wg > 0x7fffffffu) throw std::runtime_error("bary: matrices too large");
    hipEvent_t ev = ctx.prof_begin(K_BARY);
    hipLaunchKernelGGL(bary_partial_batch_k, dim3((unsigned)wg), dim3(BARY_T), 0, ctx.stream, b);
    hipLaunchKernelGGL(bary_final_batch_k, dim3((unsigned)outs), dim3(64), 0, ctx.stream, b);
    ctx.prof_end(K_BARY, ev, bytes);
    HIP_CHECK(hipGetLastError());
  }
}

// y = sum * (z^h - s^h) / (h s^h), s = GENERATOR (p3 interpolate_coset); sums indexed c * np + p
// ---- open_alpha_k: see msamd.h (OpenAlphaArgs). One workgroup of 256 threads.
namespace {
__device__ __forceinline__ u32 open_msg_word(const OpenAlphaArgs& a, u32 i) {
  // the transcript piece as 32-bit words: the 8 state words, then the opened values (c0 low, c0 high, c1 low, c1 high)
  return i < 8 ? a.state_in[i] : reinterpret_cast<const u32*>(a.opened)[i - 8];
}
constexpr u32 OPEN_MAX_ENTRIES = 512;
__device__ __forceinline__ E2 wave_sum_e2(E2 v) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    const u64 o0 = (u64)__shfl_xor((unsigned long long)v.c0, m, 64), o1 = (u64)__shfl_xor((unsigned long long)v.c1, m, 64);
    v = e2(gl_add(v.c0, o0), gl_add(v.c1, o1));
  }
  return v;
}
__global__ __launch_bounds__(256) void open_alpha_k(OpenAlphaArgs a) {
  __shared__ E2 sh_alpha;
  __shared__ E2 sh_e[OPEN_MAX_ENTRIES];  // per entry: the finishing factor, later the coefficient, later coeff * column sum
  const u32 t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // (1) finish the barycentric sums: y = sum * (z^(2^log_h) - s_pow) * dinv, into observe order. The factor is one chain of
  // log_h squarings per entry: a thread per entry computes it, then everybody scales
  for (u32 e = t; e < a.n_entries; e += 256) {
    const OpenEntry en = a.entries[e];
    sh_e[e] = e2_mul_base(e2_sub(e2_exp_pow2(a.points[en.point_id], en.log_h), e2(en.s_pow)), en.dinv);
  }
  __syncthreads();
  for (u32 e = wave; e < a.n_entries; e += 4) {
    const OpenEntry en = a.entries[e];
    const E2 scale = sh_e[e];
    for (u32 c = lane; c < en.w; c += 64) a.opened[en.out_off + c] = e2_mul(a.sums[en.sum_off + c * en.np + en.p], scale);
  }
  __threadfence_block();
  __syncthreads();
  // (2) BLAKE3 of state || values: a lane per 1024-byte chunk, then the left-full tree over the chaining values
  const u32 len = 32 + 16 * a.n_vals, nchunks = (len + 1023) / 1024;
  for (u32 ch = t; ch < nchunks; ch += 256) {
    const u32 clen = len - 1024 * ch < 1024 ? len - 1024 * ch : 1024;
    const u32 nblk = (clen + 63) / 64;
    u32 cv[8];
    b3_iv(cv);
    for (u32 b = 0; b < nblk; b++) {
      u32 m[16];
      const u32 bl = clen - 64 * b < 64 ? clen - 64 * b : 64;
#pragma unroll
      for (u32 k = 0; k < 16; k++) m[k] = 4 * k < bl ? open_msg_word(a, 256 * ch + 16 * b + k) : 0u;
      const u32 flags = (b == 0 ? (u32)B3_CHUNK_START : 0u) | (b + 1 == nblk ? (u32)B3_CHUNK_END | (nchunks == 1 ? (u32)B3_ROOT : 0u) : 0u);
      b3_compress(cv, m, ch, bl, flags);
    }
    store_digest(a.cv_scratch + ch, cv);
  }
  __threadfence_block();
  __syncthreads();
  for (u32 n = nchunks; n > 1; n = (n + 1) / 2) {  // pair adjacent values, an odd last one moves up unchanged; in place
    const u32 nn = (n + 1) / 2;
    u32 d[8];
    // node i reads 2 i and 2 i + 1 (>= i) and is written to i: passes of 256 nodes from the left, reads and writes of a pass
    // separated by a barrier, never overwrite an input a later pass still needs
    for (u32 base = 0; base < nn; base += 256) {
      const u32 i = base + t;
      const bool on = i < nn;
      if (on) {
        if (2 * i + 1 < n) {
          u32 l[8], r[8], m[16];
          load_digest(a.cv_scratch + 2 * i, l);
          load_digest(a.cv_scratch + 2 * i + 1, r);
#pragma unroll
          for (int k = 0; k < 8; k++) {
            m[k] = l[k];
            m[8 + k] = r[k];
          }
          b3_iv(d);
          b3_compress(d, m, 0, 64, B3_PARENT | (n == 2 ? (u32)B3_ROOT : 0u));
        } else {
          load_digest(a.cv_scratch + 2 * i, d);
        }
      }
      __syncthreads();
      if (on) store_digest(a.cv_scratch + i, d);
      __threadfence_block();
      __syncthreads();
    }
  }
  // (3) alpha <- sample; the input buffer afterwards is the latest digest
  if (t == 0) {
    DevChallenger s;
    u32 dg[8];
    load_digest(a.cv_scratch, dg);
    for (int k = 0; k < 8; k++) s.dg[k] = dg[k];
    s.pos = 32;
    const E2 alpha = dc_sample_ext(s);
    for (int k = 0; k < 8; k++) a.state_out[k] = s.dg[k];
    *a.alpha_out = alpha;
    sh_alpha = alpha;
  }
  __syncthreads();
  const E2 alpha = sh_alpha;
  // (4) alpha's powers, and every entry's coefficient alpha^exp * cmul (a thread per power / per entry)
  for (u32 i = t; i <= a.gw; i += 256) a.apow[i] = e2_pow(alpha, i);
  for (u32 e = t; e < a.n_entries; e += 256) {
    const OpenEntry en = a.entries[e];
    sh_e[e] = e2_mul_base(e2_pow(alpha, en.exp), en.cmul);
  }
  __threadfence_block();
  __syncthreads();
  // (5) per entry (a wave each, in turn): the column sum sum_c alpha^c y_c; lane 0 leaves the coefficient in the matrix's
  // descriptor and coeff * sum for the constants
  for (u32 e = wave; e < a.n_entries; e += 4) {
    const OpenEntry en = a.entries[e];
    if (en.mat == ~0u) continue;
    E2 part = e2(0);
    for (u32 c = lane; c < en.w; c += 64) part = e2_add(part, e2_mul(a.apow[c], a.opened[en.out_off + c]));
    part = wave_sum_e2(part);
    if (lane == 0) {
      const E2 coeff = sh_e[e];
      DeepMat& dm = a.mats[en.mat];
      dm.coeff[en.p] = coeff;
      dm.coeff7[en.p] = gl_mul(coeff.c1, GL_EXT_W);
      sh_e[e] = e2_mul(coeff, part);
    }
  }
  __syncthreads();
  // (6) the constants: K[slot] = sum over the slot's entries (field sums are exact: any order)
  for (u32 sl = t; sl < a.n_slots; sl += 256) {
    E2 k = e2(0);
    for (u32 e = 0; e < a.n_entries; e++)
      if (a.entries[e].slot == sl && a.entries[e].mat != ~0u) k = e2_add(k, sh_e[e]);
    a.K[sl] = k;
  }
}
}  // namespace

void open_alpha(Ctx& ctx, const OpenAlphaArgs& a) {
  if (!a.entries || !a.sums || !a.points || !a.state_in || !a.opened || !a.apow || !a.state_out || !a.alpha_out || !a.cv_scratch)
    throw std::runtime_error("open_alpha: null argument");
  if (a.n_entries > OPEN_MAX_ENTRIES) throw std::runtime_error("open_alpha: too many (matrix, point) pairs");
  hipLaunchKernelGGL(open_alpha_k, dim3(1), dim3(256), 0, ctx.stream, a);
  HIP_CHECK(hipGetLastError());
}

void bary_finish(const E2* sums, size_t w, unsigned log_h, const E2* zs, int npoints, E2* out /* p * w + c */) {
  size_t h = size_t(1) << log_h;
  u64 s_pow = gl_exp_pow2(GL_GEN, log_h);
  u64 dinv = gl_inv(gl_mul(s_pow, (u64)h % GL_P));
  for (int p = 0; p < npoints; p++) {
    E2 scale = e2_mul_base(e2_sub(e2_exp_pow2(zs[p], log_h), e2(s_pow)), dinv);
    for (size_t c = 0; c < w; c++) out[p * w + c] = e2_mul(sums[c * npoints + p], scale);
  }
}

void deep_reduce(Ctx& ctx, const std::vector<DeepMat>& mats, const DeepPoints& pts, size_t height, const E2* apow_dev, E2* ro,
                 const E2* apow_host, Digest* fri_leaves, const DeepMat* mats_dev, size_t row0, size_t full_height, const E2* K_dev) {
  if (pts.n > 2) throw std::runtime_error("deep_reduce: more than two opening points at one height");
  for (auto& m : mats)
    for (u32 k = 0; k < m.npoints; k++)
      if (m.npoints > 2 || m.pt[k] >= pts.n) throw std::runtime_error("deep_reduce: bad point index");
  // (the matrix list and the alpha powers stay in device memory: indexed by a loop variable inside the kernel's
  // argument block they cost more than the two small uploads save)
  (void)apow_host;
  if (!apow_dev) throw std::runtime_error("deep_reduce: alpha powers missing");
  DBuf<DeepMat> dm;
  if (!mats_dev) {  // (pcs_open uploads every height's list together with the alpha powers: one copy instead of one per height)
    dm = DBuf<DeepMat>(ctx, mats.size());
    ctx.h2d(dm.p, mats.data(), mats.size() * sizeof(DeepMat));
    mats_dev = dm.p;
  }
  DeepParams p{mats_dev, (u32)mats.size(), apow_dev, pts, ro, height, fri_leaves, 0, row0, K_dev};
  if (!full_height) full_height = height;
  if (row0 > full_height || height > full_height - row0 || (row0 & 1)) throw std::runtime_error("deep_reduce: row range outside the domain");
  if (pts.shift[0] || pts.shift[1]) {
    if (full_height & (full_height - 1)) throw std::runtime_error("deep_reduce: shifted points need a power-of-two domain");
    p.log_height = log2_strict(full_height);
  }
  double bytes = 16.0 * height * (1 + pts.n);
  for (auto& m : mats) bytes += 8.0 * m.w * height;
  hipEvent_t ev = ctx.prof_begin(K_DEEP);
  if (height < 2 || (height & 1)) throw std::runtime_error("deep_reduce: LDE height must be even");
  size_t total_w = 0;
  for (auto& m : mats) total_w += m.w;
  if (height <= 8192 && total_w >= 512 && !getenv("MSAMD_NO_DEEP_WIDE"))  // few rows, very wide: 16 column slices per row pair
    hipLaunchKernelGGL(deep_reduce_wide_k, dim3((unsigned)((height / 2 + 15) / 16)), dim3(256), 0, ctx.stream, p);
  else
    hipLaunchKernelGGL(deep_reduce_k, dim3((unsigned)((height / 2 + 255) / 256)), dim3(256), 0, ctx.stream, p);
  ctx.prof_end(K_DEEP, ev, bytes);
  HIP_CHECK(hipGetLastError());
}

void fri_fold(Ctx& ctx, const E2* cur, size_t rows, E2 beta, const E2* roll_in, E2* out, size_t row0, size_t rows_total) {
  unsigned lr = log2_strict(rows_total ? rows_total : rows);
  if (lr + 1 > TW_LOG) throw std::runtime_error("FRI layer above 2^28 is not supported");
  u64 half = gl_inv(2);
  E2 hb = e2_mul_base(beta, half);
  E2 rf = e2_sqr(beta);  // roll-in factor beta^2
  hipEvent_t ev = ctx.prof_begin(K_FRI_FOLD);
  hipLaunchKernelGGL(fri_fold_k, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx.stream, cur, rows, lr, hb, half, roll_in, rf,
                     ctx.tw0i, ctx.tw1i, out, row0);
  ctx.prof_end(K_FRI_FOLD, ev, 48.0 * rows);
  HIP_CHECK(hipGetLastError());
}

void fri_tree_build(Ctx& ctx, DTree& t, const E2* cur, size_t rows, const FriChallenge* fc, unsigned log_arity) {
  if (log_arity < 1 || log_arity > FRI_MAX_LOG_ARITY) throw std::runtime_error("FRI: round arity out of range");
  if (!t.digests.p) merkle_alloc(ctx, t, rows);  // already allocated when the fold wrote the leaf layer
  if (cur) {
    hipEvent_t ev = ctx.prof_begin(K_LEAF_HASH);
    if (log_arity == 1)
      hipLaunchKernelGGL(fri_leaf_hash_k, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx.stream, cur, rows, t.base());
    else
      hipLaunchKernelGGL(fri_leaf_hash_wide_k, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx.stream, cur, rows, log_arity, t.base());
    ctx.prof_end(K_LEAF_HASH, ev, (32.0 + 16.0 * double(1u << log_arity)) * rows);
    HIP_CHECK(hipGetLastError());
  }
  merkle_compress_plain(ctx, t, fc);
}

void fri_fold_dev(Ctx& ctx, const E2* cur, size_t rows, const FriTailRound* rec, const E2* roll_in, E2* out, Digest* next_leaves, size_t row0,
                  size_t rows_total, unsigned squarings) {
  unsigned lr = log2_strict(rows_total ? rows_total : rows);
  if (row0 + rows > (size_t(1) << lr)) throw std::runtime_error("fri_fold_dev: row range outside the layer");
  if (lr + 1 > TW_LOG) throw std::runtime_error("FRI layer above 2^28 is not supported");
  if (next_leaves && rows < 2) throw std::runtime_error("fri_fold_dev: no next layer to hash");
  const size_t threads = (rows + 1) / 2;
  hipEvent_t ev = ctx.prof_begin(K_FRI_FOLD);
  if (next_leaves)
    hipLaunchKernelGGL(fri_fold_dev_k<true>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx.stream, cur, rows, lr, rec, roll_in,
                       ctx.tw0i, ctx.tw1i, out, next_leaves, row0, (u32)squarings);
  else
    hipLaunchKernelGGL(fri_fold_dev_k<false>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx.stream, cur, rows, lr, rec,
                       roll_in, ctx.tw0i, ctx.tw1i, out, next_leaves, row0, (u32)squarings);
  ctx.prof_end(K_FRI_FOLD, ev, (next_leaves ? 64.0 : 48.0) * rows);
  HIP_CHECK(hipGetLastError());
}

// Device grinding. `input` is the challenger's pending input buffer; returns the minimal witness.
bool grind_device(Ctx& ctx, const std::vector<uint8_t>& input, unsigned bits, u64* witness_out) {
  const size_t L = input.size();
  if (L + 8 > 1024 || bits == 0 || bits > 40) return false;  // single-chunk transcripts only
  const size_t nblocks = (L + 8 + 63) / 64;
  size_t nb_pre = L / 64;
  if (nb_pre > nblocks - 1) nb_pre = nblocks - 1;
  GrindParams p;
  memset(&p, 0, sizeof(p));
  b3_iv(p.cv_mid);
  for (size_t b = 0; b < nb_pre; b++) {
    u32 m[16];
    for (int i = 0; i < 16; i++) {
      const uint8_t* q = &input[b * 64 + 4 * i];
      m[i] = (u32)q[0] | ((u32)q[1] << 8) | ((u32)q[2] << 16) | ((u32)q[3] << 24);
    }
    b3_compress(p.cv_mid, m, 0, 64, b == 0 ? B3_CHUNK_START : 0);
  }
  p.tail_len = (u32)(L - nb_pre * 64);
  if (p.tail_len + 8 > 128) return false;
  for (u32 i = 0; i < p.tail_len; i++) p.tail[i >> 2] |= (u32)input[nb_pre * 64 + i] << (8 * (i & 3));
  p.first_block = nb_pre == 0 ? 1 : 0;
  p.mask = (u64(1) << bits) - 1;
  DBuf<unsigned long long> best(ctx, 1);
  const u64 batch = u64(1) << (bits + 6 < 16 ? 16 : bits + 6 > 24 ? 24 : bits + 6);
  for (u64 w0 = 0;; w0 += batch) {
    unsigned long long init = ~0ull;
    ctx.h2d(best.p, &init, 8);
    p.w0 = w0;
    hipLaunchKernelGGL(grind_k, dim3((unsigned)(batch / 256)), dim3(256), 0, ctx.stream, p, best.p);
    HIP_CHECK(hipGetLastError());
    unsigned long long r = 0;
    ctx.d2h(&r, best.p, 8);
    if (r != ~0ull) {
      *witness_out = r;
      return true;
    }
    if (w0 + batch >= GL_P - batch) throw std::runtime_error("grind: witness space exhausted");
  }
}

// Reads the cap of tree t and, when `bits` > 0 and the transcript fits, searches the PoW witness in the same
// submission (one host synchronisation). *found = false means the caller must grind the usual way.
std::vector<Digest> cap_and_grind(Ctx& ctx, const DTree& t, const std::vector<uint8_t>& prefix, unsigned bits, bool* found,
                                  u64* witness) {
  const size_t cl = t.cap_layer();
  const size_t ncap = t.layer_len[cl];
  std::vector<Digest> cap(ncap);
  *found = false;
  const bool fits = bits > 0 && bits <= 40 && (prefix.size() % 4) == 0 && prefix.size() + 32 * ncap + 8 <= 128;
  if (!fits) {
    ctx.d2h(cap.data(), t.base() + t.layer_off[cl], ncap * sizeof(Digest));
    return cap;
  }
  GrindCapParams p;
  memset(&p, 0, sizeof(p));
  memcpy(p.prefix, prefix.data(), prefix.size());
  p.prefix_words = (u32)(prefix.size() / 4);
  p.cap_words = (u32)(8 * ncap);
  p.w0 = 0;
  p.mask = (u64(1) << bits) - 1;
  DBuf<unsigned long long> best(ctx, 1);
  HIP_CHECK(hipMemsetAsync(best.p, 0xff, 8, ctx.stream));
  const u64 batch = u64(1) << (bits + 6 < 16 ? 16 : bits + 6 > 24 ? 24 : bits + 6);
  hipLaunchKernelGGL(grind_cap_k, dim3((unsigned)(batch / 256)), dim3(256), 0, ctx.stream, p,
                     (const u32*)(t.base() + t.layer_off[cl]), best.p);
  HIP_CHECK(hipGetLastError());
  unsigned long long r = 0;
  ctx.d2h_queue(cap.data(), t.base() + t.layer_off[cl], ncap * sizeof(Digest));
  ctx.d2h(&r, best.p, 8);
  if (r != ~0ull) {
    *found = true;
    *witness = r;
  }
  return cap;
}

void fri_tail(Ctx& ctx, const E2* cur0, uint32_t len0, uint32_t n_rounds, unsigned pow_bits, uint32_t* state_dev,
              const std::vector<FriTailRoll>& rolls, Digest* tree_out, E2* layers_out, FriTailRound* rounds_dev, E2* final_dev) {
  if (len0 > 2048 || len0 < 2 || (len0 & (len0 - 1))) throw std::runtime_error("fri_tail: bad length");
  if (rolls.size() > FRI_TAIL_MAX_ROLLS) throw std::runtime_error("fri_tail: too many roll-in inputs");
  FriTailParams p;
  memset(&p, 0, sizeof(p));
  p.cur0 = cur0;
  p.len0 = len0;
  p.n_rounds = n_rounds;
  p.pow_bits = pow_bits;
  p.n_roll = (u32)rolls.size();
  p.state = state_dev;
  for (size_t i = 0; i < rolls.size(); i++) p.roll[i] = rolls[i];
  p.tree_out = tree_out;
  p.layers_out = layers_out;
  p.final_out = final_dev;
  p.rounds = rounds_dev;
  p.t0i = ctx.tw0i;
  p.t1i = ctx.tw1i;
  hipLaunchKernelGGL(fri_tail_k, dim3(1), dim3(1024), 0, ctx.stream, p);
  HIP_CHECK(hipGetLastError());
}

void fri_query_challenge(Ctx& ctx, const uint32_t* state_dev, const E2* final_dev, unsigned pow_bits, uint32_t n_queries,
                         unsigned log_max_height, u64* out_dev) {
  if (log_max_height >= 64) throw std::runtime_error("fri_query_challenge: bad height");
  hipLaunchKernelGGL(fri_query_challenge_k, dim3(1), dim3(1024), 0, ctx.stream, state_dev, final_dev, pow_bits, n_queries, log_max_height,
                     out_dev);
  HIP_CHECK(hipGetLastError());
}

void gather_queries_launch(Ctx& ctx, const std::vector<GatherSeg>& segs, GatherSeg* segs_dev, const u64* indices_dev, size_t n_queries,
                           size_t bytes_per_query, uint8_t* out_dev, unsigned owner_shift, uint32_t owner) {
  if (segs.empty() || n_queries == 0) return;
  if (owner_shift > 63) throw std::runtime_error("gather_queries: owner shift out of range");
  ctx.h2d(segs_dev, segs.data(), segs.size() * sizeof(GatherSeg));
  hipLaunchKernelGGL(gather_queries_k, dim3((unsigned)segs.size(), (unsigned)n_queries), dim3(64), 0, ctx.stream,
                     (const GatherSeg*)segs_dev, indices_dev, bytes_per_query, out_dev, (u32)owner_shift, (u32)owner);
  HIP_CHECK(hipGetLastError());
}

void gather_queries(Ctx& ctx, const std::vector<GatherSeg>& segs, const std::vector<uint64_t>& indices, size_t bytes_per_query,
                    uint8_t* host_out) {
  if (segs.empty() || indices.empty()) return;
  DBuf<GatherSeg> ds(ctx, segs.size());
  DBuf<u64> di(ctx, indices.size());
  DBuf<uint8_t> dout(ctx, bytes_per_query * indices.size());
  ctx.h2d(di.p, indices.data(), indices.size() * 8);
  gather_queries_launch(ctx, segs, ds.p, di.p, indices.size(), bytes_per_query, dout.p);
  ctx.d2h(host_out, dout.p, bytes_per_query * indices.size());
}

void gather_rows(Ctx& ctx, const std::vector<GatherReq>& reqs, uint8_t* host_out, size_t out_bytes) {
  if (reqs.empty()) return;
  DBuf<GatherReq> dr(ctx, reqs.size());
  DBuf<uint8_t> dout(ctx, out_bytes);
  ctx.h2d(dr.p, reqs.data(), reqs.size() * sizeof(GatherReq));
  hipLaunchKernelGGL(gather_k, dim3((unsigned)reqs.size()), dim3(64), 0, ctx.stream, dr.p, reqs.size(), dout.p);
  HIP_CHECK(hipGetLastError());
  ctx.d2h(host_out, dout.p, out_bytes);
}

}  // namespace msamd
