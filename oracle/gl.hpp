// ORACLE — test infrastructure only (CPU restatement of the reference's arithmetic).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything
// under oracle/. The product (multi-stark_amd/csrc) never includes or links this.
//
// Goldilocks field p = 2^64 - 2^32 + 1 and its degree-2 binomial extension X^2 = 7.
// Restates Plonky3 p3-goldilocks 0.5.1 (git e9d75614, not vendored in /root/reference):
//   GENERATOR = 7, TWO_ADICITY = 32, two_adic_generator(32) = 1753635133440165772
//   (= 7^((p-1)/2^32), checked numerically), BinomiallyExtendable<2>::W = 7.
// Reference call sites: src/types.rs:24-27 (Val, ExtVal), src/system.rs:334-349 (W recovered as X^D).
// PARITY UNPINNED at the Plonky3 boundary (no golden vectors in the reference); the values here are
// fixed by mathematics (any correct field implementation gives the same canonical elements).
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>

namespace mso {

typedef uint64_t u64;
typedef unsigned __int128 u128;

static const u64 F_P = 0xFFFFFFFF00000001ULL;
static const u64 F_EPS = 0xFFFFFFFFULL;  // 2^32 - 1 = 2^64 mod p
static const u64 F_GENERATOR = 7;
static const u64 F_TWO_ADIC_GEN_TOP = 1753635133440165772ULL;
static const u64 EXT_W = 7;        // BinomiallyExtendable<2>::W
static const unsigned EXT_D = 2;   // Challenge = BinomialExtensionField<Goldilocks, 2>
static const unsigned F_TWO_ADICITY = 32;
static const unsigned F_WIRE_BYTES = 8;  // serde: canonical u64

static inline u64 f_add(u64 a, u64 b) {
  u64 s = a + b;
  bool c = s < a;
  // a,b < p so a+b < 2p < 2^65; subtract p once if needed
  if (c || s >= F_P) s -= F_P;
  return s;
}
static inline u64 f_sub(u64 a, u64 b) { return a >= b ? a - b : a + (F_P - b); }
static inline u64 f_neg(u64 a) { return a ? F_P - a : 0; }

// 128-bit -> canonical, using 2^64 = 2^32 - 1 and 2^96 = -1 (mod p).
static inline u64 f_reduce128(u128 x) {
  u64 lo = (u64)x, hi = (u64)(x >> 64);
  u64 hi_hi = hi >> 32, hi_lo = hi & F_EPS;
  u64 t0 = lo - hi_hi;
  if (lo < hi_hi) t0 -= F_EPS;  // borrow: add p == subtract 2^32-1 (mod 2^64)
  u64 t1 = hi_lo * F_EPS;
  u64 r = t0 + t1;
  if (r < t1) r += F_EPS;  // carry: subtract p == add 2^32-1
  if (r >= F_P) r -= F_P;
  return r;
}
static inline u64 f_mul(u64 a, u64 b) { return f_reduce128((u128)a * b); }
static inline u64 f_from_u64(u64 x) { return x >= F_P ? x - F_P : x; }

static inline u64 f_pow(u64 b, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = f_mul(r, b);
    b = f_mul(b, b);
    e >>= 1;
  }
  return r;
}
static inline u64 f_inv(u64 a) { return f_pow(a, F_P - 2); }
static inline u64 f_exp_pow2(u64 a, unsigned k) {
  while (k--) a = f_mul(a, a);
  return a;
}
// generator of the multiplicative subgroup of order 2^bits
static inline u64 f_two_adic_generator(unsigned bits) { return f_exp_pow2(F_TWO_ADIC_GEN_TOP, 32 - bits); }

}  // namespace mso
