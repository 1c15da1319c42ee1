// BLAKE3 compression spread over the four lanes of a quad, for the latency-bound parts of the prover (upper Merkle
// levels, the single-workgroup FRI tail): one lane per column of the 4x4 state, the diagonal step reached by rotating
// rows 1..3 across the quad with DPP quad_perm moves. A compression is then ~300 dependent instructions per lane
// instead of ~900, i.e. a third of the latency, for ~40 % more total work - worth it exactly where most lanes
// would otherwise idle. Same function as b3_compress (b3_dev.h) with cv = IV and counter = 0, which is every
// Merkle node and every FRI leaf (single-block roots; /root/reference/src/types.rs:82-83).
#pragma once
#include "b3_dev.h"

namespace msamd {

namespace b3q {
// message schedule (same rows as the B3_ROUND lines of b3_dev.h)
constexpr unsigned char S[7][16] = {{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15},
                                    {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8},
                                    {3, 4, 10, 12, 13, 2, 7, 14, 6, 5, 9, 0, 11, 15, 8, 1},
                                    {10, 7, 12, 9, 14, 3, 13, 15, 4, 0, 11, 2, 5, 8, 1, 6},
                                    {12, 13, 9, 11, 15, 10, 14, 8, 7, 2, 5, 3, 0, 1, 6, 4},
                                    {9, 14, 11, 5, 8, 12, 15, 1, 13, 3, 0, 10, 2, 6, 4, 7},
                                    {11, 15, 5, 0, 1, 9, 8, 6, 14, 10, 2, 12, 3, 4, 7, 13}};
// word index lane c needs in round r: k = 0,1 column step (mx, my), k = 2,3 diagonal step; packed 4 bits per lane
constexpr u32 packed(int r, int k) {
  u32 p = 0;
  for (int c = 0; c < 4; c++) {
    int pos = (k < 2 ? 2 * c + k : 8 + 2 * c + (k - 2));
    p |= (u32)S[r][pos] << (4 * c);
  }
  return p;
}
__device__ __forceinline__ u32 qperm(u32 x, int ctrl) {
  // lane i of each quad reads lane perm[i]; ctrl = perm[0] | perm[1] << 2 | perm[2] << 4 | perm[3] << 6
#if !defined(__HIP_DEVICE_COMPILE__)
  return x;  // host pass of the compiler only parses device code
#else
  if (ctrl == 0x39) return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x39, 0xF, 0xF, true);
  if (ctrl == 0x4E) return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);
  return (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x93, 0xF, 0xF, true);
#endif
}
}  // namespace b3q

#define B3Q_G(mx, my)          \
  a = a + b + (mx);            \
  d = b3_rotr(d ^ a, 16);      \
  cc = cc + d;                 \
  b = b3_rotr(b ^ cc, 12);     \
  a = a + b + (my);            \
  d = b3_rotr(d ^ a, 8);       \
  cc = cc + d;                 \
  b = b3_rotr(b ^ cc, 7);

// All four lanes of a quad call this with the same `msg` (LDS, 16 words; only the first 8 are read when HALF, the
// rest of the block being zero). Lane c = lane & 3 gets digest words c and 4 + c.
template <bool HALF>
__device__ __forceinline__ void b3_quad_compress_iv(const u32* msg, u32 block_len, u32 flags, u32& h_lo, u32& h_hi) {
  const u32 c = threadIdx.x & 3;
  const u32 sh4 = 4 * c;
  u32 mw[7][4];
#pragma unroll
  for (int r = 0; r < 7; r++) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const u32 idx = (b3q::packed(r, k) >> sh4) & 15u;
      if (HALF) {
        const u32 v = msg[idx & 7u];
        mw[r][k] = idx < 8 ? v : 0u;
      } else {
        mw[r][k] = msg[idx];
      }
    }
  }
  u32 a = c == 0 ? B3_IV0 : c == 1 ? B3_IV1 : c == 2 ? B3_IV2 : B3_IV3;
  u32 b = c == 0 ? B3_IV4 : c == 1 ? B3_IV5 : c == 2 ? B3_IV6 : B3_IV7;
  u32 cc = a;
  u32 d = c == 2 ? block_len : c == 3 ? flags : 0u;  // counter = 0
#pragma unroll
  for (int r = 0; r < 7; r++) {
    B3Q_G(mw[r][0], mw[r][1])
    b = b3q::qperm(b, 0x39);
    cc = b3q::qperm(cc, 0x4E);
    d = b3q::qperm(d, 0x93);
    B3Q_G(mw[r][2], mw[r][3])
    b = b3q::qperm(b, 0x93);
    cc = b3q::qperm(cc, 0x4E);
    d = b3q::qperm(d, 0x39);
  }
  h_lo = a ^ cc;
  h_hi = b ^ d;
}

// The same with a chaining value carried from block to block (counter 0: the blocks of chunk 0 of a stream). Lane c holds
// words c and 4 + c of the chaining value in (lo, hi), in and out; `msg` = the block's 16 words (LDS).
__device__ __forceinline__ void b3_quad_compress_cv(const u32* msg, u32 block_len, u32 flags, u32& lo, u32& hi) {
  const u32 c = threadIdx.x & 3;
  const u32 sh4 = 4 * c;
  u32 mw[7][4];
#pragma unroll
  for (int r = 0; r < 7; r++) {
#pragma unroll
    for (int k = 0; k < 4; k++) mw[r][k] = msg[(b3q::packed(r, k) >> sh4) & 15u];
  }
  u32 a = lo, b = hi;
  u32 cc = c == 0 ? B3_IV0 : c == 1 ? B3_IV1 : c == 2 ? B3_IV2 : B3_IV3;
  u32 d = c == 2 ? block_len : c == 3 ? flags : 0u;  // counter = 0
#pragma unroll
  for (int r = 0; r < 7; r++) {
    B3Q_G(mw[r][0], mw[r][1])
    b = b3q::qperm(b, 0x39);
    cc = b3q::qperm(cc, 0x4E);
    d = b3q::qperm(d, 0x93);
    B3Q_G(mw[r][2], mw[r][3])
    b = b3q::qperm(b, 0x93);
    cc = b3q::qperm(cc, 0x4E);
    d = b3q::qperm(d, 0x39);
  }
  lo = a ^ cc;
  hi = b ^ d;
}

// parent of the two adjacent 32-byte digests at `children` (LDS): BLAKE3 of those 64 bytes as a root
__device__ __forceinline__ void b3_quad_parent(const u32* children, u32& h_lo, u32& h_hi) {
  b3_quad_compress_iv<false>(children, 64, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT, h_lo, h_hi);
}
// digest of one 32-byte row at `row` (LDS)
__device__ __forceinline__ void b3_quad_row32(const u32* row, u32& h_lo, u32& h_hi) {
  b3_quad_compress_iv<true>(row, 32, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT, h_lo, h_hi);
}

}  // namespace msamd
