// Microbenchmark: BabyBear Montgomery multiplication forms on gfx950 (registers only).
//   0: the compiler's lowering of bb_mul (v_mad_u64_u32 + v_mul_lo_u32 + v_mul_hi_u32 + 3)
//   1: lo * p^-1 as two shift-adds (p^-1 = 2^31 + 2^27 + 1 mod 2^32), hi(t * p) from a second v_mad_u64_u32
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../multi-stark_amd/csrc/bb_dev.h"
using namespace msbb;

__device__ __forceinline__ u32 bb_mul_v1(u32 a, u32 b) {
  u64 x, y;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(x) : "v"(a), "v"(b) : "vcc");
  const u32 lo = (u32)x, hi = (u32)(x >> 32);
  const u32 t = lo + (lo << 27) + (lo << 31);
  const u32 p = BB_P;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(y) : "v"(t), "v"(p) : "vcc");
  const u32 uh = (u32)(y >> 32);
  const u32 r = hi - uh, r2 = r + BB_P;
  return r2 < r ? r2 : r;
}

template <int OP>
__global__ __launch_bounds__(256) void k(u32* out, int iters) {
  u32 x[8];
  for (int i = 0; i < 8; i++) x[i] = (threadIdx.x * 2654435761u + i * 40503u + blockIdx.x) % BB_P;
  u32 w = 0x12345678u % BB_P;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = OP == 0 ? bb_mul(x[i], w) : bb_mul_v1(x[i], w);
    w = w + 1 < BB_P ? w + 1 : 1;
  }
  u32 s = 0;
  for (int i = 0; i < 8; i++) s ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name) {
  u32* d;
  int blocks = 256 * 8 * 2, iters = 400;
  (void)hipMalloc(&d, size_t(blocks) * 256 * 4);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 3; rep++) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  u32 h[4];
  (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("%-10s %.3f ms -> %.2f T mul/s (check %08x)\n", name, best, double(blocks) * 256 * iters * 8 / best / 1e9, h[0] ^ h[1] ^ h[2] ^ h[3]);
  (void)hipFree(d);
}

int main() {
  run<0>("compiler");
  run<1>("mad+shifts");
  return 0;
}
