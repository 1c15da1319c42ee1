// One FRI commit-phase step of the Fiat-Shamir transcript on the device: observe the round's root, grind the
// proof-of-work witness, sample beta. Restates DeterministicPow<SerializingChallenger64<Goldilocks,
// HashChallenger<u8, Blake3, 32>>> (/root/reference/src/types.rs:28-81) for the one shape the commit phase
// produces: the pending input is exactly one 32-byte digest (the state left by the previous sample), the
// commitment is a single root (cap_height 0). The host replays the same steps on its own challenger afterwards
// and rejects the proof if anything differs, so this code is an accelerator, never the authority.
#pragma once
#include "b3_dev.h"
#include "msamd.h"

namespace msamd {

struct ChallengeShared {
  u32 st[8];  // challenger input buffer (latest digest)
  u32 dg[8];  // working digest while sampling
  u32 mid[8]; // chaining value after the first transcript block (state || root)
  unsigned long long best;
  u64 wit;
  E2 beta;
};

__device__ __forceinline__ u64 be64_at(const u32* d, int pos) {
  // the challenger pops bytes from the back: the u64 built from digest bytes [pos, pos + 8) read big-endian
  return (u64)__builtin_bswap32(d[pos / 4 + 1]) | ((u64)__builtin_bswap32(d[pos / 4]) << 32);
}

// Called by every thread of an NT-thread workgroup; `root` = 8 words every thread can read (LDS). On return
// (after a barrier) s.st holds the new state, s.wit the witness (0 when pow_bits = 0) and s.beta the challenge.
template <int NT>
__device__ __forceinline__ void challenger_round(ChallengeShared& s, const u32* root, u32 pow_bits) {
  const u32 t = threadIdx.x;
  // transcript block 0 = state || root (64 bytes); its chaining value is needed by every grinding thread and by the
  // final digest, so one wave computes it once
  u64 wit = 0;
  if (pow_bits) {
    if (t < 64) {
      u32 blk0[16], mid[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        blk0[k] = s.st[k];
        blk0[8 + k] = root[k];
      }
      b3_iv(mid);
      b3_compress(mid, blk0, 0, 64, B3_CHUNK_START);
      if (t == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) s.mid[k] = mid[k];
        s.best = ~0ull;
      }
    }
    __syncthreads();
    u32 mid[8];
#pragma unroll
    for (int k = 0; k < 8; k++) mid[k] = s.mid[k];
    const u64 mask = (u64(1) << pow_bits) - 1;
    u32 cv[8];
    u64 w;
    for (u64 base = 0;; base += NT) {
      w = base + t;
      u32 m[16];
      m[0] = (u32)w;
      m[1] = (u32)(w >> 32);
#pragma unroll
      for (int k = 2; k < 16; k++) m[k] = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) cv[k] = mid[k];
      b3_compress(cv, m, 0, 8, B3_CHUNK_END | B3_ROOT);
      const u64 v = (u64)__builtin_bswap32(cv[7]) | ((u64)__builtin_bswap32(cv[6]) << 32);
      if ((v & mask) == 0) atomicMin(&s.best, (unsigned long long)w);
      __syncthreads();
      const unsigned long long b = s.best;
      __syncthreads();
      if (b != ~0ull) {
        wit = b;
        break;
      }
    }
    // the thread that tried the winning witness already holds the digest the sampling continues from
    if (w == wit) {
#pragma unroll
      for (int k = 0; k < 8; k++) s.dg[k] = cv[k];
    }
    __syncthreads();
  }
  if (t == 0) {
    int pos;
    if (pow_bits) {
      pos = 24;  // check_witness' sample_bits consumed digest bytes 24..31
    } else {
      u32 cv[8], blk0[16];
      for (int k = 0; k < 8; k++) {
        blk0[k] = s.st[k];
        blk0[8 + k] = root[k];
      }
      b3_iv(cv);
      b3_compress(cv, blk0, 0, 64, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
      for (int k = 0; k < 8; k++) s.dg[k] = cv[k];
      pos = 32;
    }
    u64 c[2];
    for (int ci = 0; ci < 2; ci++) {
      for (;;) {
        if (pos == 0) {  // output buffer exhausted: flush, i.e. digest <- BLAKE3(digest)
          u32 m[16], nv[8];
          for (int k = 0; k < 8; k++) m[k] = s.dg[k];
          for (int k = 8; k < 16; k++) m[k] = 0;
          b3_iv(nv);
          b3_compress(nv, m, 0, 32, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
          for (int k = 0; k < 8; k++) s.dg[k] = nv[k];
          pos = 32;
        }
        pos -= 8;
        const u64 v = be64_at(s.dg, pos);
        if (v < GL_P) {
          c[ci] = v;
          break;
        }
      }
    }
    for (int k = 0; k < 8; k++) s.st[k] = s.dg[k];
    s.beta = e2(c[0], c[1]);
    s.wit = wit;
  }
  __syncthreads();
}

}  // namespace msamd
