// Microbenchmark: two questions of the host-resident upload path (prover.hip, HostUpload).
// (1) A kernel that READS pinned host memory itself (narrow bytes, zero-copy over PCIe) and writes the widened 64-bit words to
//     HBM, against hipMemcpyAsync + a widening kernel: time per chunk by chunk size.
// (2) Does hipStreamWaitValue32 / hipStreamWriteValue32 work here, i.e. can a stream be made to wait for a value that is
//     written by an operation enqueued LATER on another stream (or by the host)?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);       \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

__global__ __launch_bounds__(256) void pull_widen_k(const uint8_t* __restrict__ host, size_t n, uint64_t* __restrict__ out) {
  // 16 bytes per thread per step: one 16-byte load over the link, two... sixteen 8-byte stores
  const size_t stride = size_t(gridDim.x) * blockDim.x * 16;
  for (size_t i = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) * 16; i + 16 <= n; i += stride) {
    const uint4 v = *reinterpret_cast<const uint4*>(host + i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    ulonglong2* o = reinterpret_cast<ulonglong2*>(out + i);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      o[2 * k] = make_ulonglong2(w[k] & 0xff, (w[k] >> 8) & 0xff);
      o[2 * k + 1] = make_ulonglong2((w[k] >> 16) & 0xff, w[k] >> 24);
    }
  }
}
__global__ __launch_bounds__(256) void widen_k(const uint8_t* __restrict__ in, size_t n, uint64_t* __restrict__ out) {
  const size_t stride = size_t(gridDim.x) * blockDim.x * 16;
  for (size_t i = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) * 16; i + 16 <= n; i += stride) {
    const uint4 v = *reinterpret_cast<const uint4*>(in + i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    ulonglong2* o = reinterpret_cast<ulonglong2*>(out + i);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      o[2 * k] = make_ulonglong2(w[k] & 0xff, (w[k] >> 8) & 0xff);
      o[2 * k + 1] = make_ulonglong2((w[k] >> 16) & 0xff, w[k] >> 24);
    }
  }
}
__global__ void mark_k(uint32_t* p, uint32_t v) { *p = v; }

int main() {
  const size_t total = size_t(14) << 20;  // the bench trace, narrowed: 14.7 MB
  uint8_t* h = nullptr;
  CK(hipHostMalloc((void**)&h, total, hipHostMallocDefault));
  for (size_t i = 0; i < total; i++) h[i] = (uint8_t)(i * 131);
  uint8_t* dn = nullptr;
  uint64_t* dw = nullptr;
  CK(hipMalloc((void**)&dn, total));
  CK(hipMalloc((void**)&dw, total * 8));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (size_t chunks : {1, 4, 8, 16, 32}) {
    const size_t per = total / chunks;
    for (int mode = 0; mode < 2; mode++) {
      for (unsigned grid : {64u, 256u, 1024u}) {
        if (mode == 0 && grid != 256u) continue;
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
          CK(hipEventRecord(a, s));
          for (size_t k = 0; k < chunks; k++) {
            if (mode == 0) {
              CK(hipMemcpyAsync(dn + k * per, h + k * per, per, hipMemcpyHostToDevice, s));
              hipLaunchKernelGGL(widen_k, dim3(256), dim3(256), 0, s, dn + k * per, per, dw + k * per);
            } else {
              hipLaunchKernelGGL(pull_widen_k, dim3(grid), dim3(256), 0, s, h + k * per, per, dw + k * per);
            }
          }
          CK(hipEventRecord(b, s));
          CK(hipStreamSynchronize(s));
          float ms = 0;
          CK(hipEventElapsedTime(&ms, a, b));
          if (ms < best) best = ms;
        }
        printf("%2zu chunks of %7.2f MB  %-22s grid %4u : %8.1f us  (%.1f GB/s of narrow bytes)\n", chunks, per / 1e6,
               mode == 0 ? "memcpy + widen kernel" : "kernel pulls host mem", grid, best * 1e3, total / best / 1e6);
      }
    }
  }
  // correctness of the pull
  std::vector<uint64_t> back(1024);
  CK(hipMemcpy(back.data(), dw + 12345 * 16, 1024 * 8, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < 1024; i++)
    if (back[i] != h[12345 * 16 + i]) {
      printf("pull mismatch at %zu\n", i);
      return 1;
    }
  // ---- (2) stream wait on a value written later
  int can = 0;
  hipError_t qe = hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
  printf("hipDeviceAttributeCanUseStreamWaitValue: %d (%s)\n", can, hipGetErrorString(qe));
  uint32_t* flag = nullptr;
  CK(hipHostMalloc((void**)&flag, 64, hipHostMallocDefault));
  *flag = 0;
  uint32_t* seen = nullptr;
  CK(hipHostMalloc((void**)&seen, 64, hipHostMallocDefault));
  *seen = 0;
  hipStream_t s2;
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipError_t we = hipStreamWaitValue32(s, flag, 7, hipStreamWaitValueEq, 0xffffffffu);
  printf("hipStreamWaitValue32 (host-pinned flag): %s\n", hipGetErrorString(we));
  if (we == hipSuccess) {
    hipLaunchKernelGGL(mark_k, dim3(1), dim3(1), 0, s, seen, 1u);
    std::this_thread::sleep_for(std::chrono::milliseconds(5));
    const uint32_t early = *(volatile uint32_t*)seen;
    auto t0 = std::chrono::steady_clock::now();
    hipError_t wr = hipStreamWriteValue32(s2, flag, 7, 0);
    printf("hipStreamWriteValue32 on another stream: %s\n", hipGetErrorString(wr));
    if (wr != hipSuccess) *(volatile uint32_t*)flag = 7;  // host store releases it as well
    CK(hipStreamSynchronize(s));
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("kernel behind the wait ran early? %u (must be 0); released %.1f us after the write was issued; seen %u\n", early, us, *seen);
    // and released by a plain host store
    *flag = 0;
    *seen = 0;
    CK(hipStreamWaitValue32(s, flag, 9, hipStreamWaitValueEq, 0xffffffffu));
    hipLaunchKernelGGL(mark_k, dim3(1), dim3(1), 0, s, seen, 2u);
    std::this_thread::sleep_for(std::chrono::milliseconds(2));
    const uint32_t early2 = *(volatile uint32_t*)seen;
    t0 = std::chrono::steady_clock::now();
    *(volatile uint32_t*)flag = 9;
    CK(hipStreamSynchronize(s));
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("released by a host store: early %u (must be 0), %.1f us after the store; seen %u\n", early2, us, *seen);
  }
  return 0;
}
