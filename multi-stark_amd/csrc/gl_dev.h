// Goldilocks (p = 2^64 - 2^32 + 1) and its quadratic extension X^2 = 7, for gfx950 device code and for
// the host-side scalar work of the prover (challenges, alpha powers, selectors constants).
// Field the reference instantiates: /root/reference/src/types.rs:24-27 (Val = Goldilocks, ExtVal =
// BinomialExtensionField<Val, 2>); W recovered as X^D in src/system.rs:334-349.
// All values are kept canonical (< p) in memory; 2^64 = 2^32 - 1 and 2^96 = -1 (mod p) drive the reduction.
#pragma once
#if !defined(__HIPCC_RTC__)  // hiprtc (quotient_jit.hip) has no standard library headers; its built-ins declare these
#include <cstddef>
#include <cstdint>
#endif

#if defined(__HIPCC_RTC__)
typedef unsigned long uint64_t;  // LP64, as <cstdint> has it
typedef unsigned int uint32_t;
typedef unsigned long size_t;
#define GL_HD __device__ __forceinline__
#elif defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GL_HD __host__ __device__ __forceinline__
#else
#define GL_HD inline
#endif

namespace msamd {

typedef uint64_t u64;
typedef uint32_t u32;

static constexpr u64 GL_P = 0xFFFFFFFF00000001ULL;
static constexpr u64 GL_EPS = 0xFFFFFFFFULL;
static constexpr u64 GL_GEN = 7;                          // multiplicative generator (p3 Goldilocks::GENERATOR)
static constexpr u64 GL_W32 = 1753635133440165772ULL;     // generator of the order-2^32 subgroup = 7^((p-1)/2^32)
static constexpr u64 GL_EXT_W = 7;                        // X^2 = 7

// On the device the modular add/sub/reduce are written as VCC carry chains in inline assembly: the compiler's own
// lowering of the same C (64-bit compares into SGPR pairs + v_cndmask, register-pair shuffles for v_lshl_add_u64 and
// the hazard s_nops between them) costs about twice the instructions. Every intermediate stays in VGPRs. A carry or borrow
// out of a 64-bit add / sub is folded back as +- (2^32 - 1) = "low limb -+ c, high limb +- (c xor the low limb's own
// carry-out)": the two masks live in VCC and one SGPR pair, and their xor is ONE s_xor_b64 on the SCALAR unit instead of a
// v_cndmask + a third vector add - these kernels are bound by vector-ALU issue, the scalar unit is idle (measured,
// tools/micro/gl_sgpr.hip: mul 1.38 -> 1.53 T/s, add 4.74 -> 5.80, sub 6.46 -> 7.27, bit-identical on 4.2 M operand pairs).
// The blocks clobber VCC and SCC (scalar logic writes SCC: the compiler keeps its loop conditions out of the way).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ u64 gl_pack(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
#endif

GL_HD u64 gl_sub(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  // d = a - b; on borrow add p, i.e. subtract 2^32 - 1 (mod 2^64)
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32), d0, d1;
  u64 sm;
  asm("v_sub_co_u32 %0, vcc, %3, %5\n\t"
      "v_subb_co_u32 %1, vcc, %4, %6, vcc\n\t"   // borrow B
      "v_addc_co_u32 %0, %2, 0, %0, vcc\n\t"     // + p = + 1 - 2^32: low += B (carry C2) ...
      "s_xor_b64 %2, %2, vcc\n\t"                // ... high -= B and not C2
      "v_subbrev_co_u32 %1, vcc, 0, %1, %2"
      : "=&v"(d0), "=&v"(d1), "=&s"(sm)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
      : "vcc", "scc");
  return gl_pack(d0, d1);
#else
  u64 d = a - b;
  if (a < b) d += GL_P;
  return d;
#endif
}
GL_HD u64 gl_add(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  // s = a + b (65 bits with the carry); s >= p <=> the carry, or s + (2^32 - 1) carries out; then the result is s - p
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32), s0, s1, d0, d1;
  u64 sc;
  asm("v_add_co_u32 %0, vcc, %5, %7\n\t"
      "v_addc_co_u32 %1, %4, %6, %8, vcc\n\t"    // carry C of a + b
      "v_add_co_u32 %2, vcc, -1, %0\n\t"         // t = s + (2^32 - 1) = s - p (mod 2^64): carries out iff s >= p
      "v_addc_co_u32 %3, vcc, 0, %1, vcc\n\t"
      "s_or_b64 vcc, vcc, %4\n\t"                // a + b >= p: take t
      "v_cndmask_b32 %2, %0, %2, vcc\n\t"
      "v_cndmask_b32 %3, %1, %3, vcc"
      : "=&v"(s0), "=&v"(s1), "=&v"(d0), "=&v"(d1), "=&s"(sc)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
      : "vcc", "scc");
  return gl_pack(d0, d1);
#else
  u64 s = a + b;
  if (s < a || s >= GL_P) s -= GL_P;
  return s;
#endif
}
GL_HD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }

GL_HD u64 gl_reduce128(u64 lo, u64 hi) {
  u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  u64 t0 = lo - hi_hi;
  if (lo < hi_hi) t0 -= GL_EPS;
  u64 t1 = hi_lo * GL_EPS;  // (hi_lo << 32) - hi_lo, fits in 64 bits
  u64 r = t0 + t1;
  if (r < t1) r += GL_EPS;
  if (r >= GL_P) r -= GL_P;
  return r;
}

// x = (h1:h0) * 2^64 + lo  ->  canonical x mod p = lo + ((h0 << 32) - h0) - h1: u = (h0 << 32) - h0 is formed exactly
// in two instructions, then one 64-bit add and one 64-bit sub whose carry / borrow is folded back as +/- (2^32 - 1)
// (neither fold can wrap again: lo + u - 2^64 <= 2^64 - 2^33, and a borrowed difference is >= p), then one
// conditional subtraction of p.
GL_HD u64 gl_reduce_limbs(u64 lo, u32 h0, u32 h1) {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 l0 = (u32)lo, l1 = (u32)(lo >> 32), r0, r1, t0, t1;
  u64 sm;
  asm("v_sub_co_u32 %0, vcc, 0, %7\n\t"            // u = (h0 << 32) - h0
      "v_subbrev_co_u32 %1, vcc, 0, %7, vcc\n\t"
      "v_add_co_u32 %0, vcc, %5, %0\n\t"            // A = lo + u, carry Cy
      "v_addc_co_u32 %1, vcc, %6, %1, vcc\n\t"
      "v_subbrev_co_u32 %0, %4, 0, %0, vcc\n\t"     // + (2^32 - 1): low -= Cy (borrow b2), high += Cy and not b2
      "s_xor_b64 %4, %4, vcc\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
      "v_sub_co_u32 %0, vcc, %0, %8\n\t"            // C = A - h1, borrow Bw
      "v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_addc_co_u32 %0, %4, 0, %0, vcc\n\t"        // - (2^32 - 1): low += Bw (carry c3), high -= Bw and not c3
      "s_xor_b64 %4, %4, vcc\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %1, %4\n\t"
      "v_add_co_u32 %2, vcc, -1, %0\n\t"            // C >= p  <=>  C + (2^32 - 1) carries out
      "v_addc_co_u32 %3, vcc, 0, %1, vcc\n\t"
      "v_cndmask_b32 %2, %0, %2, vcc\n\t"
      "v_cndmask_b32 %3, %1, %3, vcc"
      : "=&v"(r0), "=&v"(r1), "=&v"(t0), "=&v"(t1), "=&s"(sm)
      : "v"(l0), "v"(l1), "v"(h0), "v"(h1)
      : "vcc", "scc");
  return gl_pack(t0, t1);
#else
  return gl_reduce128(lo, ((u64)h1 << 32) | h0);
#endif
}
// the same with h1 = 0 (a 96-bit value), as produced by shifts of less than 32 bits
GL_HD u64 gl_reduce_limbs96(u64 lo, u32 h0) {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 l0 = (u32)lo, l1 = (u32)(lo >> 32), r0, r1, t0, t1;
  u64 sm;
  asm("v_sub_co_u32 %0, vcc, 0, %7\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %7, vcc\n\t"
      "v_add_co_u32 %0, vcc, %5, %0\n\t"
      "v_addc_co_u32 %1, vcc, %6, %1, vcc\n\t"
      "v_subbrev_co_u32 %0, %4, 0, %0, vcc\n\t"
      "s_xor_b64 %4, %4, vcc\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
      "v_add_co_u32 %2, vcc, -1, %0\n\t"
      "v_addc_co_u32 %3, vcc, 0, %1, vcc\n\t"
      "v_cndmask_b32 %2, %0, %2, vcc\n\t"
      "v_cndmask_b32 %3, %1, %3, vcc"
      : "=&v"(r0), "=&v"(r1), "=&v"(t0), "=&v"(t1), "=&s"(sm)
      : "v"(l0), "v"(l1), "v"(h0)
      : "vcc", "scc");
  return gl_pack(t0, t1);
#else
  return gl_reduce128(lo, (u64)h0);
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
// 64 x 64 -> 128: four v_mad_u64_u32 partial products, summed limb-wise with one carry chain
__device__ __forceinline__ void gl_mul_wide(u64 a, u64 b, u32& l0, u32& l1, u32& h0, u32& h1) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 p00, p01, p10, p11;
  asm("v_mad_u64_u32 %0, vcc, %4, %6, 0\n\t"
      "v_mad_u64_u32 %1, vcc, %4, %7, 0\n\t"
      "v_mad_u64_u32 %2, vcc, %5, %6, 0\n\t"
      "v_mad_u64_u32 %3, vcc, %5, %7, 0"
      : "=&v"(p00), "=&v"(p01), "=&v"(p10), "=&v"(p11)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
      : "vcc");
  u32 p00l = (u32)p00, p00h = (u32)(p00 >> 32), p01l = (u32)p01, p01h = (u32)(p01 >> 32);
  u32 p10l = (u32)p10, p10h = (u32)(p10 >> 32), p11l = (u32)p11, p11h = (u32)(p11 >> 32);
  asm("v_add_co_u32 %0, vcc, %3, %4\n\t"        // l1 = p00h + p01l
      "v_addc_co_u32 %1, vcc, %5, %7, vcc\n\t"  // h0 = p01h + p10h + c
      "v_addc_co_u32 %2, vcc, 0, %9, vcc\n\t"   // h1 = p11h + c
      "v_add_co_u32 %0, vcc, %0, %6\n\t"        // l1 += p10l
      "v_addc_co_u32 %1, vcc, %1, %8, vcc\n\t"  // h0 += p11l + c
      "v_addc_co_u32 %2, vcc, 0, %2, vcc"        // h1 += c
      : "=&v"(l1), "=&v"(h0), "=&v"(h1)
      : "v"(p00h), "v"(p01l), "v"(p01h), "v"(p10l), "v"(p10h), "v"(p11l), "v"(p11h)
      : "vcc");
  l0 = p00l;
}
#endif

GL_HD u64 gl_mul(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  // the four partial products, then limb sums and the reduction as ONE asm statement: the compiler pads every asm
  // boundary with an s_nop (it cannot see which hazards a block leaves open), so fewer, longer blocks are cheaper
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 p00, p01, p10, p11;
  asm("v_mad_u64_u32 %0, vcc, %4, %6, 0\n\t"
      "v_mad_u64_u32 %1, vcc, %4, %7, 0\n\t"
      "v_mad_u64_u32 %2, vcc, %5, %6, 0\n\t"
      "v_mad_u64_u32 %3, vcc, %5, %7, 0"
      : "=&v"(p00), "=&v"(p01), "=&v"(p10), "=&v"(p11)
      : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
      : "vcc");
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
      "v_subbrev_co_u32 %0, %7, 0, %0, vcc\n\t"    // + (2^32 - 1): low -= Cy (borrow b2), high += Cy and not b2
      "s_xor_b64 %7, %7, vcc\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %7\n\t"
      "v_sub_co_u32 %0, vcc, %0, %6\n\t"           // C = A - h1, borrow Bw
      "v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_addc_co_u32 %0, %7, 0, %0, vcc\n\t"       // - (2^32 - 1): low += Bw (carry c3), high -= Bw and not c3
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
#else
  unsigned __int128 x = (unsigned __int128)a * b;
  return gl_reduce128((u64)x, (u64)(x >> 64));
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
// x * 2^(64 + r), 0 <= r < 32, from the limbs (y2 : y1 : y0) of x << r:  y0 2^64 + y1 2^96 + y2 2^128 =
// ((y0 << 32) - y0) - (y2 : y1). Both operands are canonical ((2^32 - 1)^2 < p; y2 < 2^31), so ONE modular
// subtraction finishes it: 7 vector instructions where three chained subtractions took 17.
__device__ __forceinline__ u64 gl_shift_hi(u32 y0, u32 y1, u32 y2) {
  u32 t0, t1;
  u64 sm;
  asm("v_sub_co_u32 %0, vcc, 0, %3\n\t"             // T = (y0 << 32) - y0
      "v_subbrev_co_u32 %1, vcc, 0, %3, vcc\n\t"
      "v_sub_co_u32 %0, vcc, %0, %4\n\t"            // T - (y2 : y1), borrow B
      "v_subb_co_u32 %1, vcc, %1, %5, vcc\n\t"
      "v_addc_co_u32 %0, %2, 0, %0, vcc\n\t"        // + p
      "s_xor_b64 %2, %2, vcc\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %1, %2"
      : "=&v"(t0), "=&v"(t1), "=&s"(sm)
      : "v"(y0), "v"(y1), "v"(y2)
      : "vcc", "scc");
  return gl_pack(t0, t1);
}
// x * 2^(32 + r), 0 <= r < 32:  y0 2^32 + y1 2^64 + y2 2^96 = (y0 : 0) + ((y1 << 32) - y1) - y2. The sum of the first
// two is below 2 p (one carry fold), y2 < 2^31 (one borrow fold), then the conditional subtraction of p.
__device__ __forceinline__ u64 gl_shift_mid(u32 y0, u32 y1, u32 y2) {
  u32 t0, t1, r0, r1;
  u64 sm;
  asm("v_sub_co_u32 %0, vcc, 0, %6\n\t"             // T = (y1 << 32) - y1
      "v_subbrev_co_u32 %1, vcc, 0, %6, vcc\n\t"
      "v_add_co_u32 %1, vcc, %1, %5\n\t"            // + (y0 : 0): the low limb is untouched; carry Cy
      "v_subbrev_co_u32 %0, %4, 0, %0, vcc\n\t"     // + (2^32 - 1)
      "s_xor_b64 %4, %4, vcc\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, %4\n\t"
      "v_sub_co_u32 %0, vcc, %0, %7\n\t"            // - y2, borrow Bw
      "v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\t"
      "v_addc_co_u32 %0, %4, 0, %0, vcc\n\t"        // - (2^32 - 1)
      "s_xor_b64 %4, %4, vcc\n\t"
      "v_subbrev_co_u32 %1, vcc, 0, %1, %4\n\t"
      "v_add_co_u32 %2, vcc, -1, %0\n\t"            // >= p ?
      "v_addc_co_u32 %3, vcc, 0, %1, vcc\n\t"
      "v_cndmask_b32 %2, %0, %2, vcc\n\t"
      "v_cndmask_b32 %3, %1, %3, vcc"
      : "=&v"(t0), "=&v"(t1), "=&v"(r0), "=&v"(r1), "=&s"(sm)
      : "v"(y0), "v"(y1), "v"(y2)
      : "vcc", "scc");
  return gl_pack(r0, r1);
}
#endif

// x * 2^k mod p for 0 <= k < 96 by shifts (2 has order 192 and 2^96 = -1: every root of unity of order <= 64 is a
// signed power of two, so the innermost NTT stages need no multiplier). Meant for k known at compile time.
GL_HD u64 gl_mul_2exp(u64 x, unsigned k) {
  if (k == 0) return x;
  if (k <= 32) {
    u64 lo = x << k, hi = x >> (64 - k);
    return gl_reduce_limbs96(lo, (u32)hi);
  }
#if defined(__HIP_DEVICE_COMPILE__)
  {
    // x << r as three limbs (r = k mod 32, k > 32), then the limb arithmetic of the exponent's 32-bit block
    const unsigned r = k & 31;
    const u64 lo = x << r;
    const u32 y2 = r ? ((u32)(x >> 32) >> (32 - r)) : 0u;
    return k < 64 ? gl_shift_mid((u32)lo, (u32)(lo >> 32), y2) : gl_shift_hi((u32)lo, (u32)(lo >> 32), y2);
  }
#else
  if (k < 64) {
    u64 lo = x << k, hi = x >> (64 - k);
    return gl_reduce_limbs(lo, (u32)hi, (u32)(hi >> 32));
  }
  // 64 <= k < 96: (hi * 2^64 + lo) * 2^64 with (hi:lo) = x << (k - 64), hi < 2^32:
  //   hi * 2^128 + l1 * 2^96 + l0 * 2^64 = -(hi << 32) - l1 + (l0 << 32) - l0
  unsigned s = k - 64;
  u64 lo = x << s;
  u64 hi = s ? (x >> (64 - s)) : 0;
  u32 l0 = (u32)lo, l1 = (u32)(lo >> 32);
  u64 t = gl_sub(((u64)l0 << 32), (u64)l0);   // (l0 << 32) - l0, both canonical
  t = gl_sub(t, (u64)l1);
  return gl_sub(t, hi << 32);
#endif
}
GL_HD u64 gl_sqr(u64 a) { return gl_mul(a, a); }

// Lazy dot-product accumulator: sums 64x64-bit products in 160 bits (lo, hi, carry count) and reduces once.
// A term costs four multiply-adds and five adds instead of a full multiplication plus a modular addition.
struct GlAcc {
  u64 lo, hi;
  u32 c;
};
GL_HD void acc_init(GlAcc& a) {
  a.lo = 0;
  a.hi = 0;
  a.c = 0;
}
GL_HD void acc_mad(GlAcc& a, u64 x, u64 y) {
#if defined(__HIP_DEVICE_COMPILE__)
  u32 l0, l1, h0, h1;
  gl_mul_wide(x, y, l0, l1, h0, h1);
  u32 a0 = (u32)a.lo, a1 = (u32)(a.lo >> 32), a2 = (u32)a.hi, a3 = (u32)(a.hi >> 32), c = a.c;
  asm("v_add_co_u32 %0, vcc, %0, %5\n\t"
      "v_addc_co_u32 %1, vcc, %1, %6, vcc\n\t"
      "v_addc_co_u32 %2, vcc, %2, %7, vcc\n\t"
      "v_addc_co_u32 %3, vcc, %3, %8, vcc\n\t"
      "v_addc_co_u32 %4, vcc, 0, %4, vcc"
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c)
      : "v"(l0), "v"(l1), "v"(h0), "v"(h1)
      : "vcc");
  a.lo = gl_pack(a0, a1);
  a.hi = gl_pack(a2, a3);
  a.c = c;
#else
  unsigned __int128 pr = (unsigned __int128)x * y;
  u64 plo = (u64)pr, phi = (u64)(pr >> 64);
  u64 lo = a.lo + plo;
  u64 c1 = lo < plo ? 1u : 0u;
  u64 hi = a.hi + phi;
  u32 c2 = hi < phi ? 1u : 0u;
  u64 hi2 = hi + c1;
  c2 += hi2 < c1 ? 1u : 0u;
  a.lo = lo;
  a.hi = hi2;
  a.c += c2;
#endif
}
// value = c * 2^128 + hi * 2^64 + lo, with 2^128 = -2^32 (mod p); c stays far below 2^31 for any realistic sum
GL_HD u64 acc_reduce(const GlAcc& a) {
  u64 r = gl_reduce_limbs(a.lo, (u32)a.hi, (u32)(a.hi >> 32));
  return gl_sub(r, (u64)a.c << 32);
}
// The same sum with one 64-bit accumulator per partial product weight (x0 y0 | x0 y1 + x1 y0 | x1 y1) and a carry counter
// each: a term is four v_mad_u64_u32 that add straight into their accumulator plus four carry increments - 12 issue slots
// against 19 for acc_mad (whose limb sums need a 6-instruction carry chain before the 5-instruction accumulation) - at the
// price of 9 registers per accumulator instead of 5. For the long dot products of the opening (barycentric sums, DEEP
// column sums).
struct GlAccS {
  u64 lo, mid, hi;
  u32 clo, cmid, chi;
};
GL_HD void accs_init(GlAccS& a) {
  a.lo = a.mid = a.hi = 0;
  a.clo = a.cmid = a.chi = 0;
}
GL_HD void accs_mad(GlAccS& a, u64 x, u64 y) {
#if defined(__HIP_DEVICE_COMPILE__)
  const u32 x0 = (u32)x, x1 = (u32)(x >> 32), y0 = (u32)y, y1 = (u32)(y >> 32);
  asm("v_mad_u64_u32 %0, vcc, %6, %8, %0\n\t"
      "v_addc_co_u32 %3, vcc, 0, %3, vcc\n\t"
      "v_mad_u64_u32 %1, vcc, %6, %9, %1\n\t"
      "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
      "v_mad_u64_u32 %1, vcc, %7, %8, %1\n\t"
      "v_addc_co_u32 %4, vcc, 0, %4, vcc\n\t"
      "v_mad_u64_u32 %2, vcc, %7, %9, %2\n\t"
      "v_addc_co_u32 %5, vcc, 0, %5, vcc"
      : "+v"(a.lo), "+v"(a.mid), "+v"(a.hi), "+v"(a.clo), "+v"(a.cmid), "+v"(a.chi)
      : "v"(x0), "v"(x1), "v"(y0), "v"(y1)
      : "vcc");
#else
  const u64 x0 = (u32)x, x1 = x >> 32, y0 = (u32)y, y1 = y >> 32;
  u64 t = x0 * y0;
  a.lo += t;
  a.clo += a.lo < t;
  t = x0 * y1;
  a.mid += t;
  a.cmid += a.mid < t;
  t = x1 * y0;
  a.mid += t;
  a.cmid += a.mid < t;
  t = x1 * y1;
  a.hi += t;
  a.chi += a.hi < t;
#endif
}
// value = lo + mid 2^32 + (hi + clo) 2^64 + cmid 2^96 + chi 2^128: gathered into the 160-bit form of GlAcc with plain
// integer carries, then reduced once (the counters stay far below 2^31 for any realistic sum)
GL_HD u64 accs_reduce(const GlAccS& a) {
  GlAcc t;
  const u64 m = a.mid << 32;
  t.lo = a.lo + m;
  const u64 c = t.lo < m;
  u64 h = a.hi + c;
  u32 cc = h < c;
  const u64 add2 = (a.mid >> 32) + a.clo;
  h += add2;
  cc += h < add2;
  const u64 add3 = (u64)a.cmid << 32;
  h += add3;
  cc += h < add3;
  t.hi = h;
  t.c = a.chi + cc;
  return acc_reduce(t);
}
// multiply by a small constant c < 2^32: the high word is < 2^32 so only the 2^64 = 2^32 - 1 fold is needed
GL_HD u64 gl_mul_small(u64 a, u32 c) {
#if defined(__HIP_DEVICE_COMPILE__)
  u64 lo = a * (u64)c;
  u64 hi = __umul64hi(a, (u64)c);
#else
  unsigned __int128 x = (unsigned __int128)a * c;
  u64 lo = (u64)x, hi = (u64)(x >> 64);
#endif
  u64 t1 = hi * GL_EPS;
  u64 r = lo + t1;
  if (r < t1) r += GL_EPS;
  if (r >= GL_P) r -= GL_P;
  return r;
}

GL_HD u64 gl_pow(u64 b, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, b);
    b = gl_mul(b, b);
    e >>= 1;
  }
  return r;
}
GL_HD u64 gl_exp_pow2(u64 a, unsigned k) {
  for (unsigned i = 0; i < k; i++) a = gl_mul(a, a);
  return a;
}
// a^(p-2) by an addition chain on the exponent 0xFFFFFFFEFFFFFFFF (72 multiplications)
GL_HD u64 gl_inv(u64 a) {
  u64 t2 = gl_mul(gl_sqr(a), a);                 // 2^2 - 1
  u64 t3 = gl_mul(gl_sqr(t2), a);                // 2^3 - 1
  u64 t6 = gl_mul(gl_exp_pow2(t3, 3), t3);       // 2^6 - 1
  u64 t12 = gl_mul(gl_exp_pow2(t6, 6), t6);      // 2^12 - 1
  u64 t24 = gl_mul(gl_exp_pow2(t12, 12), t12);   // 2^24 - 1
  u64 t30 = gl_mul(gl_exp_pow2(t24, 6), t6);     // 2^30 - 1
  u64 t31 = gl_mul(gl_sqr(t30), a);              // 2^31 - 1
  u64 t32 = gl_mul(gl_sqr(t31), a);              // 2^32 - 1
  // exponent = (2^31 - 1) * 2^33 + (2^32 - 1)
  u64 r = gl_exp_pow2(t31, 33);
  return gl_mul(r, t32);
}
static constexpr unsigned GL_TWO_ADICITY = 32;
// generator of the subgroup of order 2^bits, bits <= 32 (callers bound bits; beyond the 2-adicity there is no such root
// and the unsigned difference below would wrap into billions of squarings: answer with 0, which no caller can mistake
// for a root of unity)
GL_HD u64 gl_two_adic_generator(unsigned bits) { return bits > GL_TWO_ADICITY ? 0 : gl_exp_pow2(GL_W32, GL_TWO_ADICITY - bits); }

struct E2 {
  u64 c0, c1;
};
GL_HD E2 e2(u64 a, u64 b = 0) {
  E2 r;
  r.c0 = a;
  r.c1 = b;
  return r;
}
GL_HD E2 e2_add(E2 a, E2 b) { return e2(gl_add(a.c0, b.c0), gl_add(a.c1, b.c1)); }
GL_HD E2 e2_sub(E2 a, E2 b) { return e2(gl_sub(a.c0, b.c0), gl_sub(a.c1, b.c1)); }
GL_HD E2 e2_neg(E2 a) { return e2(gl_neg(a.c0), gl_neg(a.c1)); }
GL_HD E2 e2_mul(E2 a, E2 b) {
  // Karatsuba: 3 base multiplications + one multiplication by the small constant 7
  u64 v0 = gl_mul(a.c0, b.c0), v1 = gl_mul(a.c1, b.c1);
  u64 cross = gl_sub(gl_sub(gl_mul(gl_add(a.c0, a.c1), gl_add(b.c0, b.c1)), v0), v1);
  return e2(gl_add(v0, gl_mul_small(v1, (u32)GL_EXT_W)), cross);
}
GL_HD E2 e2_sqr(E2 a) {
  u64 v0 = gl_sqr(a.c0), v1 = gl_sqr(a.c1);
  u64 m = gl_mul(a.c0, a.c1);
  return e2(gl_add(v0, gl_mul_small(v1, (u32)GL_EXT_W)), gl_add(m, m));
}
GL_HD E2 e2_mul_base(E2 a, u64 b) { return e2(gl_mul(a.c0, b), gl_mul(a.c1, b)); }
// In-place inverses of a[0 .. count) (count <= K, all non-zero) with ONE base-field inversion: 1/a = conj(a) / N(a),
// N(a) = a0^2 - 7 a1^2 in the base field, and the norms are inverted together by Montgomery's trick. Per element
// that is 7 base multiplications and 8-byte prefix products, against 3 extension multiplications (about 10 base
// ones) and 16-byte prefixes for the same trick done in the extension field.
template <int K>
GL_HD void e2_batch_inverse(E2 (&a)[K], int count) {
  u64 nrm[K], pre[K];
  u64 acc = 1;
#pragma unroll
  for (int t = 0; t < K; t++) {
    if (t < count) {
      nrm[t] = gl_sub(gl_sqr(a[t].c0), gl_mul_small(gl_sqr(a[t].c1), (u32)GL_EXT_W));
      pre[t] = acc;
      acc = gl_mul(acc, nrm[t]);
    }
  }
  u64 inv = gl_inv(acc);
#pragma unroll
  for (int t = K - 1; t >= 0; t--) {
    if (t < count) {
      const u64 ni = gl_mul(inv, pre[t]);
      inv = gl_mul(inv, nrm[t]);
      a[t] = e2(gl_mul(a[t].c0, ni), gl_mul(gl_neg(a[t].c1), ni));
    }
  }
}

GL_HD E2 e2_inv(E2 a) {
  u64 norm = gl_sub(gl_sqr(a.c0), gl_mul_small(gl_sqr(a.c1), (u32)GL_EXT_W));
  u64 ni = gl_inv(norm);
  return e2(gl_mul(a.c0, ni), gl_mul(gl_neg(a.c1), ni));
}
GL_HD bool e2_is_zero(E2 a) { return (a.c0 | a.c1) == 0; }
GL_HD E2 e2_pow(E2 b, u64 e) {
  E2 r = e2(1);
  while (e) {
    if (e & 1) r = e2_mul(r, b);
    b = e2_sqr(b);
    e >>= 1;
  }
  return r;
}
GL_HD E2 e2_exp_pow2(E2 a, unsigned k) {
  for (unsigned i = 0; i < k; i++) a = e2_sqr(a);
  return a;
}

GL_HD u32 bitrev32(u32 x, unsigned bits) {
  if (bits == 0) return 0;
#if defined(__HIP_DEVICE_COMPILE__)
  return __brev(x) >> (32 - bits);
#else
  u32 r = 0;
  for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
#endif
}
GL_HD u64 bitrev64(u64 x, unsigned bits) {
  u64 r = 0;
  for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1ull) << (bits - 1 - i);
  return r;
}
inline unsigned log2_strict(size_t n) {
  unsigned l = 0;
  while ((size_t(1) << l) < n) l++;
  return l;
}

}  // namespace msamd
