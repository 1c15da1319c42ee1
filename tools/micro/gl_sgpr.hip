// Microbenchmark + bit-exactness check of carry folds that go through the SCALAR unit: the fold of a carry / borrow into a
// 64-bit value (+- (2^32 - 1)) is "low limb +- c, high limb -+ (c xor carry-out)", and the xor of two lane masks is one
// s_xor_b64 on the scalar ALU instead of a v_cndmask on the vector ALU that binds these kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>
#include "../../multi-stark_amd/csrc/gl_dev.h"
using namespace msamd;

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ u64 sub2(u64 a, u64 b) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32), d0, d1;
  u64 sm;
  asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
      "v_subb_co_u32 %1, vcc, %4, %6, vcc\n\t"         // borrow B
      "v_addc_co_u32 %0, %2, 0, %0, vcc\n\t"           // d0 += B, carry C2
      "s_xor_b64 %2, %2, vcc\n\t"                      // lanes whose high limb loses one: B and not C2
      "v_subbrev_co_u32 %1, vcc, 0, %1, %2"
      : "=&v"(d0), "=&v"(d1), "=&s"(sm)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
      : "vcc", "scc");
  return gl_pack(d0, d1);
}
__device__ __forceinline__ u64 add2(u64 a, u64 b) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32), s0, s1, t0, t1;
  u64 sc;
  asm("v_add_co_u32 %0, vcc, %5, %7\n\t"
      "v_addc_co_u32 %1, %4, %6, %8, vcc\n\t"          // carry C of a + b
      "v_add_co_u32 %2, vcc, -1, %0\n\t"               // t = s + (2^32 - 1): carries out iff s >= p
      "v_addc_co_u32 %3, vcc, 0, %1, vcc\n\t"
      "s_or_b64 vcc, vcc, %4\n\t"
      "v_cndmask_b32 %2, %0, %2, vcc\n\t"
      "v_cndmask_b32 %3, %1, %3, vcc"
      : "=&v"(s0), "=&v"(s1), "=&v"(t0), "=&v"(t1), "=&s"(sc)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
      : "vcc", "scc");
  return gl_pack(t0, t1);
}
__device__ __forceinline__ u64 mul2(u64 a, u64 b) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 p00, p01, p10, p11;
  asm("v_mad_u64_u32 %0, vcc, %4, %6, 0\n\t"
      "v_mad_u64_u32 %1, vcc, %4, %7, 0\n\t"
      "v_mad_u64_u32 %2, vcc, %5, %6, 0\n\t"
      "v_mad_u64_u32 %3, vcc, %5, %7, 0"
      : "=&v"(p00), "=&v"(p01), "=&v"(p10), "=&v"(p11)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
      : "vcc", "scc");
  u32 p00l = (u32)p00, p00h = (u32)(p00 >> 32), p01l = (u32)p01, p01h = (u32)(p01 >> 32);
  u32 p10l = (u32)p10, p10h = (u32)(p10 >> 32), p11l = (u32)p11, p11h = (u32)(p11 >> 32);
  u32 r0, r1, t0, t1, l1, h0, h1;
  u64 sm;
  asm("v_add_co_u32 %4, vcc, %9, %10\n\t"          // l1 = p00h + p01l
      "v_addc_co_u32 %5, vcc, %11, %13, vcc\n\t"   // h0 = p01h + p10h + c
      "v_addc_co_u32 %6, vcc, 0, %15, vcc\n\t"     // h1 = p11h + c
      "v_add_co_u32 %4, vcc, %4, %12\n\t"          // l1 += p10l
      "v_addc_co_u32 %5, vcc, %5, %14, vcc\n\t"    // h0 += p11l + c
      "v_addc_co_u32 %6, vcc, 0, %6, vcc\n\t"      // h1 += c
      "v_sub_co_u32 %0, vcc, 0, %5\n\t"            // u = (h0 << 32) - h0
      "v_subbrev_co_u32 %1, vcc, 0, %5, vcc\n\t"
      "v_add_co_u32 %0, vcc, %8, %0\n\t"           // A = (l1:l0) + u, carry Cy
      "v_addc_co_u32 %1, vcc, %4, %1, vcc\n\t"
      "v_subbrev_co_u32 %0, %7, 0, %0, vcc\n\t"    // fold + (2^32 - 1): low -= Cy (borrow b2), high += Cy xor b2
      "s_xor_b64 %7, %7, vcc\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %7\n\t"
      "v_sub_co_u32 %0, vcc, %0, %6\n\t"           // C = A - h1, borrow Bw
      "v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_addc_co_u32 %0, %7, 0, %0, vcc\n\t"       // fold - (2^32 - 1): low += Bw (carry c3), high -= Bw xor c3
      "s_xor_b64 %7, %7, vcc\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %1, %7\n\t"
      "v_add_co_u32 %2, vcc, -1, %0\n\t"           // C >= p  <=>  C + (2^32 - 1) carries out
      "v_addc_co_u32 %3, vcc, 0, %1, vcc\n\t"
      "v_cndmask_b32 %2, %0, %2, vcc\n\t"
      "v_cndmask_b32 %3, %1, %3, vcc"
      : "=&v"(r0), "=&v"(r1), "=&v"(t0), "=&v"(t1), "=&v"(l1), "=&v"(h0), "=&v"(h1), "=&s"(sm)
      : "v"(p00l), "v"(p00h), "v"(p01l), "v"(p01h), "v"(p10l), "v"(p10h), "v"(p11l), "v"(p11h)
      : "vcc", "scc");
  return gl_pack(t0, t1);
}

#else
__device__ u64 sub2(u64 a, u64 b);
__device__ u64 add2(u64 a, u64 b);
__device__ u64 mul2(u64 a, u64 b);
#endif

template <int OP, bool NEW>
__global__ __launch_bounds__(256) void k(u64* out, int iters) {
  u64 x[8];
  for (int i = 0; i < 8; i++) x[i] = (threadIdx.x * 0x9E3779B97F4A7C15ULL + i * 0x123456789ULL + blockIdx.x) % GL_P;
  u64 w = 0x1234567890ABCDEFULL % GL_P;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) x[i] = NEW ? mul2(x[i], w) : gl_mul(x[i], w);
      if (OP == 1) x[i] = NEW ? add2(x[i], w) : gl_add(x[i], w);
      if (OP == 2) x[i] = NEW ? sub2(x[i], w) : gl_sub(x[i], w);
    }
    if (OP == 3) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        u64 a = x[i], b = x[i + 4];
        x[i] = NEW ? add2(a, b) : gl_add(a, b);
        x[i + 4] = NEW ? mul2(sub2(a, b), w) : gl_mul(gl_sub(a, b), w);
      }
    }
  }
  u64 s = 0;
  for (int i = 0; i < 8; i++) s ^= x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void check_k(const u64* a, const u64* b, size_t n, unsigned long long* bad) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  if (add2(a[i], b[i]) != gl_add(a[i], b[i])) atomicAdd(bad, 1ull);
  if (sub2(a[i], b[i]) != gl_sub(a[i], b[i])) atomicAdd(bad + 1, 1ull);
  if (mul2(a[i], b[i]) != gl_mul(a[i], b[i])) atomicAdd(bad + 2, 1ull);
}

// gl_mul_2exp(x, k) for every k < 96 against the general multiplication by 2^k mod p
template <unsigned K>
__device__ __forceinline__ void check_exp(u64 x, const u64* pow2, unsigned long long* bad) {
  if (gl_mul_2exp(x, K) != gl_mul(x, pow2[K])) atomicAdd(bad + 3, 1ull);
  if constexpr (K + 1 < 96) check_exp<K + 1>(x, pow2, bad);
}
__global__ void check_exp_k(const u64* a, size_t n, const u64* pow2, unsigned long long* bad) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (i < n) check_exp<0>(a[i], pow2, bad);
}

template <int OP, bool NEW>
float run(double ops_per_iter, const char* name) {
  u64* d;
  int blocks = 256 * 8 * 2, iters = 400;
  (void)hipMalloc(&d, size_t(blocks) * 256 * 8);
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 3; rep++) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<OP, NEW>), dim3(blocks), dim3(256), 0, 0, d, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  double n = double(blocks) * 256 * iters * ops_per_iter;
  printf("%-10s %-4s %.3f ms -> %.2f T ops/s\n", name, NEW ? "new" : "old", best, n / best / 1e9);
  (void)hipFree(d);
  return best;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  printf("start\n");
  (void)hipSetDevice(0);
  (void)hipFree(nullptr);
  printf("runtime up\n");
  // bit-exactness on random and edge operands (all canonical)
  const u64 edge[] = {0, 1, 2, GL_P - 1, GL_P - 2, 0xFFFFFFFFull, 0x100000000ull, 0xFFFFFFFF00000000ull, 0xFFFFFFFEFFFFFFFFull, 0x7FFFFFFFFFFFFFFFull,
                      0x8000000000000000ull, 0xFFFFFFFE00000001ull, 0x00000000FFFFFFFEull, 0xFFFFFFFF00000000ull - 1};
  const size_t ne = sizeof(edge) / sizeof(edge[0]);
  std::vector<u64> ha, hb;
  for (size_t i = 0; i < ne; i++)
    for (size_t j = 0; j < ne; j++) ha.push_back(edge[i] % GL_P), hb.push_back(edge[j] % GL_P);
  u64 s = 0x243F6A8885A308D3ull;
  auto rnd = [&]() {
    s ^= s << 13, s ^= s >> 7, s ^= s << 17;
    return s;
  };
  for (int i = 0; i < (1 << 22); i++) {
    u64 x = rnd() % GL_P, y = rnd() % GL_P;
    if ((i & 7) == 0) y = GL_P - x;                   // sums that hit p exactly
    if ((i & 7) == 1) y = (GL_P - x + (rnd() & 3)) % GL_P;
    if ((i & 7) == 2) x = (rnd() & 0xFFFFFFFFull), y = GL_P - 1 - (rnd() & 0xFFFF);
    if ((i & 7) == 3) x |= 0xFFFFFFFF00000000ull, x %= GL_P;
    ha.push_back(x), hb.push_back(y);
  }
  const size_t n = ha.size();
  printf("operands ready\n");
  u64 *da, *db;
  unsigned long long* dbad;
  (void)hipMalloc(&da, n * 8), (void)hipMalloc(&db, n * 8), (void)hipMalloc(&dbad, 32);
  (void)hipMemcpy(da, ha.data(), n * 8, hipMemcpyHostToDevice), (void)hipMemcpy(db, hb.data(), n * 8, hipMemcpyHostToDevice);
  (void)hipMemset(dbad, 0, 32);
  hipLaunchKernelGGL(check_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, da, db, n, dbad);
  {
    std::vector<u64> pw(96);
    unsigned __int128 v = 1;
    for (int k = 0; k < 96; k++) pw[k] = (u64)v, v = (v * 2) % GL_P;
    u64* dp;
    (void)hipMalloc(&dp, 96 * 8);
    (void)hipMemcpy(dp, pw.data(), 96 * 8, hipMemcpyHostToDevice);
    const size_t ne2 = std::min<size_t>(n, 1 << 18);
    hipLaunchKernelGGL(check_exp_k, dim3((unsigned)((ne2 + 255) / 256)), dim3(256), 0, 0, da, ne2, dp, dbad);
  }
  unsigned long long bad[4];
  (void)hipMemcpy(bad, dbad, 32, hipMemcpyDeviceToHost);
  printf("gl_mul_2exp for k = 0..95 on 2^18 operands: %llu mismatches\n", bad[3]);
  (void)hipDeviceSynchronize();
  printf("check kernel done: %s\n", hipGetErrorString(hipGetLastError()));
  printf("checked %zu operand pairs: mismatches add %llu sub %llu mul %llu\n", n, bad[0], bad[1], bad[2]);
  run<0, false>(8, "mul"), run<0, true>(8, "mul");
  run<1, false>(8, "add"), run<1, true>(8, "add");
  run<2, false>(8, "sub"), run<2, true>(8, "sub");
  run<3, false>(4, "butterfly"), run<3, true>(4, "butterfly");
  return (bad[0] | bad[1] | bad[2] | bad[3]) ? 1 : 0;
}
