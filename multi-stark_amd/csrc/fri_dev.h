// FRI fold of one row pair on the device, shared by the fold kernels (open.hip) and the fused round kernel (hash.hip).
// p3-fri fold_matrix for arity 2 (restated): out[i] = (lo + hi)/2 + beta (lo - hi) / (2 x_i), x_i = w_{2R}^{bitrev(i)}
// over the R rows of the folded layer, plus beta^2 * roll[i] when a shorter reduced opening joins at this length.
#pragma once
#include "msamd.h"

namespace msamd {

struct FoldArgs {
  const E2* cur;      // layer being folded: 2 * rows elements
  const E2* roll;     // nullable: reduced opening of length rows
  E2* out;            // folded layer: rows elements
  const u64 *t0i, *t1i;
  u32 log_rows;
  u32 pad;
};

// x / 2 mod p for canonical x (p is odd: an odd x borrows p first)
__device__ __forceinline__ u64 gl_half(u64 x) { return (x >> 1) + ((x & 1) ? 0x7FFFFFFF80000001ULL : 0ULL); }

// (1/2 + pw) lo + (1/2 - pw) hi, written as (lo + hi) / 2 + pw (lo - hi): one extension product instead of two, the
// halving a shift (the value is the same field element, hence the same canonical words)
__device__ __forceinline__ E2 fri_fold_value(E2 lo, E2 hi, E2 pw) {
  const E2 s = e2_add(lo, hi);
  return e2_add(e2(gl_half(s.c0), gl_half(s.c1)), e2_mul(pw, e2_sub(lo, hi)));
}

// out[i] for one row i of the folded layer; hb = beta / 2, rf = beta^2
__device__ __forceinline__ E2 fri_fold_one(const FoldArgs& f, size_t i, E2 hb, E2 rf) {
  const u32 e = bitrev32((u32)i, f.log_rows) << (TW_LOG - f.log_rows - 1);
  const u64 gp = gl_mul(f.t1i[e >> TW_HALF], f.t0i[e & ((1u << TW_HALF) - 1)]);
  const E2 pw = e2_mul_base(hb, gp);
  E2 r = fri_fold_value(f.cur[2 * i], f.cur[2 * i + 1], pw);
  if (f.roll) r = e2_add(r, e2_mul(rf, f.roll[i]));
  return r;
}

// message block of the ExtensionMmcs leaf (o0, o1): 32 bytes, the rest of the block zero
__device__ __forceinline__ void fri_row_block(E2 o0, E2 o1, u32 m[16]) {
  m[0] = (u32)o0.c0;
  m[1] = (u32)(o0.c0 >> 32);
  m[2] = (u32)o0.c1;
  m[3] = (u32)(o0.c1 >> 32);
  m[4] = (u32)o1.c0;
  m[5] = (u32)(o1.c0 >> 32);
  m[6] = (u32)o1.c1;
  m[7] = (u32)(o1.c1 >> 32);
#pragma unroll
  for (int k = 8; k < 16; k++) m[k] = 0;
}

}  // namespace msamd
