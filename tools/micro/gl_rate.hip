// Microbenchmark: Goldilocks mul / add / butterfly throughput on gfx950 (registers only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../multi-stark_amd/csrc/gl_dev.h"
using namespace msamd;

template <int OP>
__global__ __launch_bounds__(256) void k(u64* out, int iters) {
  u64 x[8];
  for (int i = 0; i < 8; i++) x[i] = (threadIdx.x * 0x9E3779B97F4A7C15ULL + i * 0x123456789ULL + blockIdx.x) % GL_P;
  u64 w = 0x1234567890ABCDEFULL % GL_P;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) x[i] = gl_mul(x[i], w);
      if (OP == 1) x[i] = gl_add(x[i], w);
      if (OP == 2) x[i] = gl_sub(x[i], w);
    }
    if (OP == 3) {  // 4 DIF butterflies
#pragma unroll
      for (int i = 0; i < 4; i++) {
        u64 a = x[i], b = x[i + 4];
        x[i] = gl_add(a, b);
        x[i + 4] = gl_mul(gl_sub(a, b), w);
      }
    }
    if (OP == 4) {
#pragma unroll
      for (int i = 0; i < 8; i++) x[i] = gl_mul_2exp(x[i], 36);
    }
  }
  u64 s = 0;
  for (int i = 0; i < 8; i++) s ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, double ops_per_iter) {
  u64* d;
  int blocks = 256 * 8 * 2, iters = 400;
  (void)hipMalloc(&d, size_t(blocks) * 256 * 8);
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
  double n = double(blocks) * 256 * iters * ops_per_iter;
  printf("%-12s %.3f ms -> %.2f T ops/s\n", name, best, n / best / 1e9);
  (void)hipFree(d);
}

int main() {
  run<0>("mul", 8);
  run<1>("add", 8);
  run<2>("sub", 8);
  run<3>("butterfly", 4);
  run<4>("mul_2exp36", 8);
  return 0;
}
