// Microbenchmark: dot-product accumulation forms on gfx950 (registers only): acc_mad (limb sums + 160-bit accumulate),
// accs_mad (one 64-bit accumulator per partial product, carries through VCC), and the same with each carry in an SGPR pair
// of its own (no VCC dependency between neighbouring instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../multi-stark_amd/csrc/gl_dev.h"
using namespace msamd;

__device__ __forceinline__ void accs_mad_sgpr(GlAccS& a, u64 x, u64 y) {
  const u32 x0 = (u32)x, x1 = (u32)(x >> 32), y0 = (u32)y, y1 = (u32)(y >> 32);
  u64 c0, c1, c2, c3;
  asm("v_mad_u64_u32 %0, %6, %10, %12, %0\n\t"
      "v_mad_u64_u32 %2, %8, %11, %13, %2\n\t"
      "v_addc_co_u32 %3, %6, 0, %3, %6\n\t"
      "v_mad_u64_u32 %1, %7, %10, %13, %1\n\t"
      "v_addc_co_u32 %5, %8, 0, %5, %8\n\t"
      "v_addc_co_u32 %4, %7, 0, %4, %7\n\t"
      "v_mad_u64_u32 %1, %9, %11, %12, %1\n\t"
      "s_nop 0\n\t"
      "v_addc_co_u32 %4, %9, 0, %4, %9"
      : "+v"(a.lo), "+v"(a.mid), "+v"(a.hi), "+v"(a.clo), "+v"(a.cmid), "+v"(a.chi), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
      : "v"(x0), "v"(x1), "v"(y0), "v"(y1));
}

template <int OP>
__global__ __launch_bounds__(256) void k(u64* out, int iters) {
  u64 x[4], y[4];
  for (int i = 0; i < 4; i++) {
    x[i] = (threadIdx.x * 0x9E3779B97F4A7C15ULL + i * 0x123456789ULL + blockIdx.x) % GL_P;
    y[i] = (threadIdx.x * 0xD1B54A32D192ED03ULL + i * 0x987654321ULL + blockIdx.x) % GL_P;
  }
  GlAcc a[4];
  GlAccS s[4];
  for (int i = 0; i < 4; i++) {
    acc_init(a[i]);
    accs_init(s[i]);
  }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (OP == 0) acc_mad(a[i], x[i], y[(i + it) & 3]);
      if (OP == 1) accs_mad(s[i], x[i], y[(i + it) & 3]);
      if (OP == 2) accs_mad_sgpr(s[i], x[i], y[(i + it) & 3]);
    }
  }
  u64 r = 0;
  for (int i = 0; i < 4; i++) r ^= OP == 0 ? acc_reduce(a[i]) : accs_reduce(s[i]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
void run(const char* name) {
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
  double n = double(blocks) * 256 * iters * 4;
  u64 h[4];
  (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  printf("%-14s %.3f ms -> %.2f T mad/s (check %016llx)\n", name, best, n / best / 1e9, (unsigned long long)(h[0] ^ h[1] ^ h[2] ^ h[3]));
  (void)hipFree(d);
}

int main() {
  run<0>("acc_mad");
  run<1>("accs_mad_vcc");
  run<2>("accs_mad_sgpr");
  return 0;
}
