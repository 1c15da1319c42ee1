// Microbenchmark: BLAKE3 compressions per second on gfx950, registers only (no memory traffic).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../multi-stark_amd/csrc/gl_dev.h"
using namespace msamd;
#ifndef ROT_VARIANT
#define ROT_VARIANT 0
#endif
__device__ __forceinline__ u32 rot(u32 x, int n) {
#if ROT_VARIANT == 0
  return __builtin_amdgcn_alignbit(x, x, n);
#elif ROT_VARIANT == 1
  if (n == 16) return __builtin_amdgcn_perm(x, x, 0x01000302u);
  if (n == 8) return __builtin_amdgcn_perm(x, x, 0x00030201u);
  return __builtin_amdgcn_alignbit(x, x, n);
#else
  return (x >> n) | (x << (32 - n));
#endif
}
#define G(a, b, c, d, mx, my) a = a + b + (mx); d = rot(d ^ a, 16); c = c + d; b = rot(b ^ c, 12); a = a + b + (my); d = rot(d ^ a, 8); c = c + d; b = rot(b ^ c, 7);
#define RND(m, s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
  G(v0, v4, v8, v12, m[s0], m[s1]) G(v1, v5, v9, v13, m[s2], m[s3]) G(v2, v6, v10, v14, m[s4], m[s5]) G(v3, v7, v11, v15, m[s6], m[s7]) \
  G(v0, v5, v10, v15, m[s8], m[s9]) G(v1, v6, v11, v12, m[s10], m[s11]) G(v2, v7, v8, v13, m[s12], m[s13]) G(v3, v4, v9, v14, m[s14], m[s15])
__device__ __forceinline__ void b3_compress(u32 cv[8], const u32 m[16], u64 counter, u32 bl, u32 fl) {
  u32 v0 = cv[0], v1 = cv[1], v2 = cv[2], v3 = cv[3], v4 = cv[4], v5 = cv[5], v6 = cv[6], v7 = cv[7];
  u32 v8 = 0x6A09E667u, v9 = 0xBB67AE85u, v10 = 0x3C6EF372u, v11 = 0xA54FF53Au, v12 = (u32)counter, v13 = (u32)(counter >> 32), v14 = bl, v15 = fl;
  RND(m, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  RND(m, 2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8)
  RND(m, 3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1)
  RND(m, 10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6)
  RND(m, 12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4)
  RND(m, 9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7)
  RND(m, 11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13)
  cv[0] = v0 ^ v8; cv[1] = v1 ^ v9; cv[2] = v2 ^ v10; cv[3] = v3 ^ v11; cv[4] = v4 ^ v12; cv[5] = v5 ^ v13; cv[6] = v6 ^ v14; cv[7] = v7 ^ v15;
}

template <int VARIANT>
__global__ __launch_bounds__(256) void k(u32* out, int iters) {
  u32 cv[8], m[16];
  for (int i = 0; i < 8; i++) cv[i] = threadIdx.x * 31 + i + blockIdx.x;
  for (int i = 0; i < 16; i++) m[i] = threadIdx.x * 17 + i;
  for (int it = 0; it < iters; it++) {
    b3_compress(cv, m, it, 64, 11);
    m[it & 15] ^= cv[0];
  }
  u32 s = 0;
  for (int i = 0; i < 8; i++) s ^= cv[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  u32* d;
  int blocks = 256 * 8 * 4, iters = 200;
  hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(a);
    hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double n = double(blocks) * 256 * iters;
    printf("compressions %.3g in %.3f ms -> %.1f G/s\n", n, ms, n / ms / 1e6);
  }
  return 0;
}
