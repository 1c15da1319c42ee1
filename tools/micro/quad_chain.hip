// Microbenchmark: latency of dependent BLAKE3 compressions on ONE workgroup (the shape of the upper Merkle levels and
// of the FRI rounds): one-lane compression vs the quad form (b3_quad.h), with and without barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../multi-stark_amd/csrc/b3_quad.h"
using namespace msamd;

// chain of `iters` dependent compressions, no barrier: variant 0 one lane per compression, 1 quad
template <int V>
__global__ __launch_bounds__(1024) void chain_k(u32* out, int iters) {
  __shared__ __attribute__((aligned(16))) u32 sh[1024 * 16];
  const u32 t = threadIdx.x;
  for (int i = 0; i < 16; i++) sh[t * 16 + i] = t * 31 + i;
  __syncthreads();
  if (V == 0) {
    u32 l[8], r[8], d[8];
    for (int i = 0; i < 8; i++) l[i] = sh[t * 16 + i], r[i] = sh[t * 16 + 8 + i];
    for (int it = 0; it < iters; it++) {
      b3_compress_pair_root(l, r, d);
      for (int i = 0; i < 8; i++) l[i] = d[i];
    }
    u32 s = 0;
    for (int i = 0; i < 8; i++) s ^= l[i];
    out[t] = s;
  } else {
    const u32 q = t >> 2, c = t & 3;
    u32 lo = 0, hi = 0;
    for (int it = 0; it < iters; it++) {
      b3_quad_parent(sh + 64 * q, lo, hi);  // each quad owns 4 rows of 16 words: no cross-quad hazard
      sh[64 * q + c] = lo;
      sh[64 * q + 4 + c] = hi;
    }
    out[t] = lo ^ hi;
  }
}

// `reps` trees of 1024 leaves -> root in one workgroup: variant 0 = quads, two passes unrolled (tree_tail_k's form);
// 1 = quads, pass loop not unrolled; 2 = one lane per node
template <int V>
__global__ __launch_bounds__(1024) void tree_k(u32* out, int reps) {
  __shared__ __attribute__((aligned(16))) u32 sh[1024 * 8];
  const u32 t = threadIdx.x, quad = t >> 2, c = t & 3;
  u32 acc = 0;
  for (int rep = 0; rep < reps; rep++) {
    for (int i = 0; i < 8; i++) sh[t * 8 + i] = t * 31 + i + rep + acc;
    __syncthreads();
    for (u32 n = 512; n >= 1; n >>= 1) {
      if (V == 0) {
        u32 lo[2], hi[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
          const u32 q = quad + 256 * p;
          if (q < n) b3_quad_parent(sh + 16 * q, lo[p], hi[p]);
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 2; p++) {
          const u32 q = quad + 256 * p;
          if (q < n) sh[8 * q + c] = lo[p], sh[8 * q + 4 + c] = hi[p];
        }
        __syncthreads();
      } else if (V == 1) {
        u32 lo[2] = {0, 0}, hi[2] = {0, 0};
#pragma unroll 1
        for (int p = 0; p < 2; p++) {
          const u32 q = quad + 256 * p;
          if (q < n) {
            u32 a, b;
            b3_quad_parent(sh + 16 * q, a, b);
            if (p == 0) lo[0] = a, hi[0] = b; else lo[1] = a, hi[1] = b;
          }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 2; p++) {
          const u32 q = quad + 256 * p;
          if (q < n) sh[8 * q + c] = lo[p], sh[8 * q + 4 + c] = hi[p];
        }
        __syncthreads();
      } else {
        u32 d[8];
        if (t < n) {
          u32 l[8], r[8];
          for (int i = 0; i < 8; i++) l[i] = sh[16 * t + i], r[i] = sh[16 * t + 8 + i];
          b3_compress_pair_root(l, r, d);
        }
        __syncthreads();
        if (t < n)
          for (int i = 0; i < 8; i++) sh[8 * t + i] = d[i];
        __syncthreads();
      }
    }
    acc ^= sh[0];
    __syncthreads();
  }
  if (t == 0) out[0] = acc;
}

template <class F>
float timeit(F f) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  u32* d;
  hipMalloc(&d, 1 << 20);
  const int iters = 2000, reps = 200;
  for (int threads : {64, 256, 1024}) {
    float a = timeit([&] { hipLaunchKernelGGL(chain_k<0>, dim3(1), dim3(threads), 0, 0, d, iters); });
    float b = timeit([&] { hipLaunchKernelGGL(chain_k<1>, dim3(1), dim3(threads), 0, 0, d, iters); });
    printf("chain, %4d threads: one-lane %.3f us per compression, quad %.3f us\n", threads, 1e3 * a / iters, 1e3 * b / iters);
  }
  float t0 = timeit([&] { hipLaunchKernelGGL(tree_k<0>, dim3(1), dim3(1024), 0, 0, d, reps); });
  float t1 = timeit([&] { hipLaunchKernelGGL(tree_k<1>, dim3(1), dim3(1024), 0, 0, d, reps); });
  float t2 = timeit([&] { hipLaunchKernelGGL(tree_k<2>, dim3(1), dim3(1024), 0, 0, d, reps); });
  printf("tree 1024 -> 1 (10 levels): quads unrolled %.2f us, quads looped %.2f us, one lane per node %.2f us\n", 1e3 * t0 / reps,
         1e3 * t1 / reps, 1e3 * t2 / reps);
  return 0;
}
