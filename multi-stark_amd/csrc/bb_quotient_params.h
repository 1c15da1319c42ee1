// Argument block of the BabyBear quotient kernels: the interpreter (bb_kernels.hip quotient_k) and the per-circuit
// straight-line kernel that quotient_jit.hip prints and compiles with hiprtc (which is why this header includes nothing
// but bb_dev.h). quotient_values, /root/reference/src/prover.rs:756-962.
#pragma once
#include "bb_dev.h"

namespace msbb {

struct QuotArgs {
  const u32 *kind, *na, *nb;
  unsigned n_nodes;
  const u32* zeros;
  unsigned n_zeros;
  const u32 *lk_mult, *lk_off, *lk_args;
  unsigned L;
  const u32 *pre, *s1, *s2;
  size_t pre_ld, s1_ld, s2_ld;
  unsigned log_n, log_q;
  u32 publics[16], delta[4];
  const E4* apow;  // constraint_count weights: alpha^(count - 1 - j) for constraint j (src/prover.rs:798-808)
  u32* out;
  size_t out_ld;
  u32* scratch;    // the interpreter's slot file (nodes x rows words); unused by the compiled kernel
  size_t row0, rows, stride;
  u32 g, w_big, gn_inv, g_pow_n, w_q;
};

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ size_t quot_bitrev(size_t x, unsigned bits) { return bits ? (size_t)(__brevll((unsigned long long)x) >> (64 - bits)) : 0; }
#endif

}  // namespace msbb
