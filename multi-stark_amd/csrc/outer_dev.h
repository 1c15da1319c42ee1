// Device side of the outer transcript's sampling, shared by outer.hip and by the claims digest's last kernel (hash.hip), which
// samples beta and gamma in the launch that produces the digest. See outer.hip for what is restated (src/types.rs:28-29).
#pragma once
#include "b3_dev.h"
#include "lookup_params.h"
#include "msamd.h"

namespace msamd {

struct DevChallenger {
  u32 dg[8];  // latest digest = the challenger's input buffer while its output buffer is being consumed
  int pos;    // bytes of dg not yet sampled (popped from the back)
};

__device__ __forceinline__ u64 dc_be64(const u32* d, int pos) {
  return (u64)__builtin_bswap32(d[pos / 4 + 1]) | ((u64)__builtin_bswap32(d[pos / 4]) << 32);
}

// BLAKE3 of `len` bytes (a multiple of 4, 0 < len <= 1024) held as words in `msg`, zero-padded to the end of the last block
static __device__ inline void dc_hash_chunk(const u32* msg, u32 len, u32 out[8]) {
  u32 cv[8];
  b3_iv(cv);
  const u32 nblk = (len + 63) / 64;
  for (u32 b = 0; b < nblk; b++) {
    u32 m[16];
#pragma unroll
    for (int k = 0; k < 16; k++) m[k] = msg[16 * b + k];
    const bool last = b + 1 == nblk;
    b3_compress(cv, m, 0, last ? len - 64 * b : 64u, (b == 0 ? (u32)B3_CHUNK_START : 0u) | (last ? (u32)(B3_CHUNK_END | B3_ROOT) : 0u));
  }
#pragma unroll
  for (int k = 0; k < 8; k++) out[k] = cv[k];
}

static __device__ inline u64 dc_sample_base(DevChallenger& s) {
  for (;;) {
    if (s.pos == 0) {  // output buffer exhausted: digest <- BLAKE3(digest)
      u32 m[16], nv[8];
      for (int k = 0; k < 8; k++) m[k] = s.dg[k];
      for (int k = 8; k < 16; k++) m[k] = 0;
      b3_iv(nv);
      b3_compress(nv, m, 0, 32, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
      for (int k = 0; k < 8; k++) s.dg[k] = nv[k];
      s.pos = 32;
    }
    s.pos -= 8;
    const u64 v = dc_be64(s.dg, s.pos);
    if (v < GL_P) return v;
  }
}
static __device__ inline E2 dc_sample_ext(DevChallenger& s) {
  const u64 a = dc_sample_base(s);
  const u64 b = dc_sample_base(s);
  return e2(a, b);
}
__device__ __forceinline__ void put_ext(u32* w, E2 e) {
  w[0] = (u32)e.c0;
  w[1] = (u32)(e.c0 >> 32);
  w[2] = (u32)e.c1;
  w[3] = (u32)(e.c1 >> 32);
}

// digest = BLAKE3 of everything up to and including the claims. beta <- sample, observe; gamma <- sample, observe.
// state_out (12 words) = the challenger's input afterwards: latest digest || gamma.
// (called by the 64 threads of a wave-sized workgroup; `digest` = 8 words every thread can read)
__device__ __forceinline__ void outer_beta_gamma_step(const u32* digest, ChallengeBG* __restrict__ bg, u32* __restrict__ state_out) {
  __shared__ E2 sh_gamma;
  const u32 t = threadIdx.x;
  if (t == 0) {
    DevChallenger s;
    for (int k = 0; k < 8; k++) s.dg[k] = digest[k];
    s.pos = 32;
    const E2 beta = dc_sample_ext(s);
    u32 msg[16];
    for (int k = 0; k < 8; k++) msg[k] = s.dg[k];
    put_ext(msg + 8, beta);
    for (int k = 12; k < 16; k++) msg[k] = 0;
    dc_hash_chunk(msg, 48, s.dg);
    s.pos = 32;
    const E2 gamma = dc_sample_ext(s);
    for (int k = 0; k < 8; k++) state_out[k] = s.dg[k];
    put_ext(state_out + 8, gamma);
    bg->beta = beta;
    bg->gamma = gamma;
    bg->gp.n = MAX_GPOW;
    sh_gamma = gamma;
  }
  __syncthreads();
  static_assert(MAX_GPOW <= 64, "one thread per power");
  if (t < MAX_GPOW) bg->gp.g[t] = e2_pow(sh_gamma, t);
}


}  // namespace msamd
