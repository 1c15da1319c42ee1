// Digest moves and one level of a Merkle sub-tree held in LDS, shared by the tree kernels (hash.hip) and the FRI tail
// (open.hip). Dependent BLAKE3 compressions are latency-bound: tools/micro/quad_chain.hip measures 1.17 us for a
// one-lane compression and 0.6 us on a quad (b3_quad.h), but a quad pass over all 1024 threads of a workgroup costs
// 3.2 us - so a level with 128 nodes or more runs one lane per node and a smaller one runs on quads.
#pragma once
#include "b3_dev.h"
#include "b3_quad.h"
#include "msamd.h"

namespace msamd {

__device__ __forceinline__ void store_digest(Digest* p, const u32 cv[8]) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
  q[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}
__device__ __forceinline__ void load_digest(const Digest* p, u32 cv[8]) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 a = q[0], b = q[1];
  cv[0] = a.x;
  cv[1] = a.y;
  cv[2] = a.z;
  cv[3] = a.w;
  cv[4] = b.x;
  cv[5] = b.y;
  cv[6] = b.z;
  cv[7] = b.w;
}

__device__ __forceinline__ void lds_store_digest(u32* sh, u32 idx, const u32 cv[8]) {
  uint4* q = reinterpret_cast<uint4*>(sh + idx * 8);
  q[0] = make_uint4(cv[0], cv[1], cv[2], cv[3]);
  q[1] = make_uint4(cv[4], cv[5], cv[6], cv[7]);
}
__device__ __forceinline__ void lds_load_digest(const u32* sh, u32 idx, u32 cv[8]) {
  const uint4* q = reinterpret_cast<const uint4*>(sh + idx * 8);
  uint4 a = q[0], b = q[1];
  cv[0] = a.x;
  cv[1] = a.y;
  cv[2] = a.z;
  cv[3] = a.w;
  cv[4] = b.x;
  cv[5] = b.y;
  cv[6] = b.z;
  cv[7] = b.w;
}

// n nodes (n <= 1024) from the 2n digests at sh (digest i at sh + 8 i): results to sh[0 .. n) and to gout[0 .. n).
// Called by every thread of a 1024-thread workgroup; ends with a barrier.
__device__ __forceinline__ void tree_level_plain(u32* sh, u32 n, Digest* gout) {
  const u32 t = threadIdx.x;
  if (n >= 128) {
    u32 d[8];
    const bool act = t < n;
    if (act) {
      u32 l[8], r[8];
      lds_load_digest(sh, 2 * t, l);
      lds_load_digest(sh, 2 * t + 1, r);
      b3_compress_pair_root(l, r, d);
    }
    __syncthreads();
    if (act) {
      lds_store_digest(sh, t, d);
      store_digest(gout + t, d);
    }
    __syncthreads();
  } else {
    const u32 q = t >> 2, c = t & 3;
    u32 lo = 0, hi = 0;
    if (q < n) b3_quad_parent(sh + 16 * q, lo, hi);
    __syncthreads();
    if (q < n) {
      u32* out = reinterpret_cast<u32*>(gout);
      sh[8 * q + c] = lo;
      sh[8 * q + 4 + c] = hi;
      out[8 * q + c] = lo;
      out[8 * q + 4 + c] = hi;
    }
    __syncthreads();
  }
}

// Every remaining level from n <= 16 nodes down to the root (2n digests at sh, as above), by the FIRST WAVE alone: its
// quads hand each level over with a wave barrier, where tree_level_plain would stop all sixteen waves of the workgroup
// twice per level. Results to sh and, level after level, to gout (n + n/2 + ... + 1 digests). Called by every thread;
// ends with a barrier.
__device__ __forceinline__ void tree_levels_first_wave(u32* sh, u32 n, Digest* gout) {
  const u32 t = threadIdx.x;
  if (t < 64) {
    const u32 q = t >> 2, c = t & 3;
    u32* out = reinterpret_cast<u32*>(gout);
    for (; n >= 1; n >>= 1) {
      u32 lo = 0, hi = 0;
      if (q < n) b3_quad_parent(sh + 16 * q, lo, hi);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (q < n) {
        sh[8 * q + c] = lo;
        sh[8 * q + 4 + c] = hi;
        out[8 * q + c] = lo;
        out[8 * q + 4 + c] = hi;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      out += 8 * n;
    }
  }
  __syncthreads();
}

}  // namespace msamd
