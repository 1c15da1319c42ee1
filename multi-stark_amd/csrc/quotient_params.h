// Kernel argument block of the quotient kernels, shared by quotient.hip (the node-program interpreter) and by the
// per-circuit kernels that quotient_jit.hip generates and compiles with hiprtc at System::new (this header is on
// their include path, so both sides agree on the layout by construction).
#pragma once
#include "gl_dev.h"

namespace msamd {

#if defined(__HIPCC_RTC__)
static constexpr unsigned TW_LOG = 28;   // must equal msamd.h (checked by a static_assert in quotient.hip)
static constexpr unsigned TW_HALF = 14;
#endif

struct QParams {
  const u64 *pre, *s1, *s2;
  size_t pre_h, s1_h, s2_h;
  unsigned log_n, log_q;
  u64 publics[8];
  u64 delta_scaled[2];
  u64 g_inv;          // inverse of the trace-domain generator
  const u64* zh;      // q entries: Z_H on the coset, x^n - 1
  const u64* zh_inv;  // q entries
  const E2* alpha_rev;  // constraint_count reversed powers
  const uint32_t* code;
  const u64* consts;
  const uint32_t* zero_slots;
  const uint32_t* lookup_slots;
  uint32_t n_instr, n_zeros, n_lookups, n_slots;
  const u64* t0;
  const u64* t1;
  u64* out;
  u64* scratch;       // global slot storage (when !LDS)
  size_t row0, rows;  // batch of storage rows handled by this launch
  E2 gpow[32];        // gamma^0 .. gamma^31: fingerprints as unreduced base x ext dot products
  // the same three tables inside the argument block (no upload, scalar loads with literal offsets): what the
  // per-circuit kernels read when the circuit has at most QP_INLINE_ALPHA constraints and quotient degree <= 8
  u64 zh_in[8], zh_inv_in[8];
  E2 alpha_rev_in[64];
};
constexpr unsigned QP_INLINE_ALPHA = 64;

__device__ __forceinline__ void mul2(u64 a0, u64 a1, u64 b0, u64 b1, u64& c0, u64& c1) {
  u64 v0 = gl_mul(a0, b0), v1 = gl_mul(a1, b1);
  u64 cross = gl_sub(gl_sub(gl_mul(gl_add(a0, a1), gl_add(b0, b1)), v0), v1);
  c0 = gl_add(v0, gl_mul_small(v1, (u32)GL_EXT_W));
  c1 = cross;
}

}  // namespace msamd
