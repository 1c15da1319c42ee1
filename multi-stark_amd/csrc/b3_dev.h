// BLAKE3 (unkeyed hash mode) for device kernels and for the host-side Fiat-Shamir challenger.
// Replaces p3 SerializingHasher<Blake3> / CompressionFunctionFromHasher<Blake3,2,32> / HashChallenger's hasher
// as instantiated in /root/reference/src/types.rs:28-29,82-83,199-207.
// The 16-word state lives entirely in registers; message words are addressed with compile-time indices
// (the permutation schedule is unrolled), so nothing spills to scratch.
#pragma once
#include "gl_dev.h"

namespace msamd {

static constexpr u32 B3_IV0 = 0x6A09E667u, B3_IV1 = 0xBB67AE85u, B3_IV2 = 0x3C6EF372u, B3_IV3 = 0xA54FF53Au,
                     B3_IV4 = 0x510E527Fu, B3_IV5 = 0x9B05688Cu, B3_IV6 = 0x1F83D9ABu, B3_IV7 = 0x5BE0CD19u;
enum : u32 { B3_CHUNK_START = 1, B3_CHUNK_END = 2, B3_PARENT = 4, B3_ROOT = 8 };

GL_HD u32 b3_rotr(u32 x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_alignbit(x, x, n);
#else
  return (x >> n) | (x << (32 - n));
#endif
}

#define B3_G(a, b, c, d, mx, my) \
  a = a + b + (mx);              \
  d = b3_rotr(d ^ a, 16);        \
  c = c + d;                     \
  b = b3_rotr(b ^ c, 12);        \
  a = a + b + (my);              \
  d = b3_rotr(d ^ a, 8);         \
  c = c + d;                     \
  b = b3_rotr(b ^ c, 7);

// one round with message words given by the (compile-time) schedule s0..s15
#define B3_ROUND(m, s0, s1, s2, s3, s4, s5, s6, s7, s8, s9, s10, s11, s12, s13, s14, s15) \
  B3_G(v0, v4, v8, v12, m[s0], m[s1])                                                       \
  B3_G(v1, v5, v9, v13, m[s2], m[s3])                                                       \
  B3_G(v2, v6, v10, v14, m[s4], m[s5])                                                      \
  B3_G(v3, v7, v11, v15, m[s6], m[s7])                                                      \
  B3_G(v0, v5, v10, v15, m[s8], m[s9])                                                      \
  B3_G(v1, v6, v11, v12, m[s10], m[s11])                                                    \
  B3_G(v2, v7, v8, v13, m[s12], m[s13])                                                     \
  B3_G(v3, v4, v9, v14, m[s14], m[s15])

// cv (8 words, in/out), m (16 message words), counter, block_len, flags. Writes the new chaining value.
GL_HD void b3_compress(u32 cv[8], const u32 m[16], u64 counter, u32 block_len, u32 flags) {
  u32 v0 = cv[0], v1 = cv[1], v2 = cv[2], v3 = cv[3], v4 = cv[4], v5 = cv[5], v6 = cv[6], v7 = cv[7];
  u32 v8 = B3_IV0, v9 = B3_IV1, v10 = B3_IV2, v11 = B3_IV3;
  u32 v12 = (u32)counter, v13 = (u32)(counter >> 32), v14 = block_len, v15 = flags;
  B3_ROUND(m, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)
  B3_ROUND(m, 2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8)
  B3_ROUND(m, 3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1)
  B3_ROUND(m, 10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6)
  B3_ROUND(m, 12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4)
  B3_ROUND(m, 9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7)
  B3_ROUND(m, 11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13)
  cv[0] = v0 ^ v8;
  cv[1] = v1 ^ v9;
  cv[2] = v2 ^ v10;
  cv[3] = v3 ^ v11;
  cv[4] = v4 ^ v12;
  cv[5] = v5 ^ v13;
  cv[6] = v6 ^ v14;
  cv[7] = v7 ^ v15;
}

GL_HD void b3_iv(u32 cv[8]) {
  cv[0] = B3_IV0;
  cv[1] = B3_IV1;
  cv[2] = B3_IV2;
  cv[3] = B3_IV3;
  cv[4] = B3_IV4;
  cv[5] = B3_IV5;
  cv[6] = B3_IV6;
  cv[7] = B3_IV7;
}

// parent / 2-to-1 compression node: BLAKE3 of exactly 64 bytes (left digest || right digest) as a root
GL_HD void b3_compress_pair_root(const u32 l[8], const u32 r[8], u32 out[8]) {
  u32 m[16];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    m[i] = l[i];
    m[8 + i] = r[i];
  }
  b3_iv(out);
  b3_compress(out, m, 0, 64, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
}

}  // namespace msamd
