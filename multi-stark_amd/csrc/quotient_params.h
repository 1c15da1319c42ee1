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

// Everything of a quotient launch that depends on the transcript's challenges lives in DEVICE memory, not in the argument
// block: the host writes it (h2d) when it knows the challenges, or the device transcript does (outer.hip) - the launch is
// then queued before alpha has reached the host. The reversed alpha powers follow in an array of their own (QParams::alpha_rev).
struct QDyn {
  u64 publics[8];       // beta, gamma, acc_in, acc_out as (c0, c1) pairs (src/lookup.rs:78-84)
  u64 delta_scaled[2];  // (acc_out - acc_in) / (n g)
  E2 gpow[32];          // gamma^0 .. gamma^31: fingerprints as unreduced base x ext dot products
};

struct QParams {
  const u64 *pre, *s1, *s2;
  size_t pre_h, s1_h, s2_h;
  unsigned log_n, log_q;
  const QDyn* dyn;
  u64 g_inv;          // inverse of the trace-domain generator
  const u64* zh;      // q entries: Z_H on the coset, x^n - 1
  const u64* zh_inv;  // q entries
  const E2* alpha_rev;  // constraint_count reversed powers (device memory)
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
  // the Z_H tables inside the argument block (no upload, scalar loads with literal offsets): what the per-circuit kernels read
  // when the circuit has at most QP_INLINE_ALPHA constraints and quotient degree <= 8
  u64 zh_in[8], zh_inv_in[8];
};
constexpr unsigned QP_INLINE_ALPHA = 64;

// the challenge-dependent block of one circuit from the values themselves (host side of quotient_eval; the device transcript
// of outer.hip runs the same arithmetic in a kernel): publics8 = [beta, gamma, acc_in, acc_out], k reversed alpha powers
GL_HD void quotient_dyn_fill(QDyn& d, E2* alpha_rev, size_t k, const u64 publics8[8], E2 alpha, u64 inj_norm) {
  for (int i = 0; i < 8; i++) d.publics[i] = publics8[i];
  d.delta_scaled[0] = gl_mul(gl_sub(publics8[6], publics8[4]), inj_norm);
  d.delta_scaled[1] = gl_mul(gl_sub(publics8[7], publics8[5]), inj_norm);
  E2 g = e2(1);
  const E2 gam = e2(publics8[2], publics8[3]);
  for (int i = 0; i < 32; i++) {
    d.gpow[i] = g;
    g = e2_mul(g, gam);
  }
  E2 ap = e2(1);
  for (size_t i = 0; i < k; i++) {
    alpha_rev[k - 1 - i] = ap;
    ap = e2_mul(ap, alpha);
  }
}

__device__ __forceinline__ void mul2(u64 a0, u64 a1, u64 b0, u64 b1, u64& c0, u64& c1) {
  u64 v0 = gl_mul(a0, b0), v1 = gl_mul(a1, b1);
  u64 cross = gl_sub(gl_sub(gl_mul(gl_add(a0, a1), gl_add(b0, b1)), v0), v1);
  c0 = gl_add(v0, gl_mul_small(v1, (u32)GL_EXT_W));
  c1 = cross;
}

}  // namespace msamd
