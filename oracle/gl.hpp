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

static const u64 GL_P = 0xFFFFFFFF00000001ULL;
static const u64 GL_EPS = 0xFFFFFFFFULL;  // 2^32 - 1 = 2^64 mod p
static const u64 GL_GENERATOR = 7;
static const u64 GL_TWO_ADIC_GEN_32 = 1753635133440165772ULL;
static const u64 GL_EXT_W = 7;

static inline u64 gl_add(u64 a, u64 b) {
  u64 s = a + b;
  bool c = s < a;
  // a,b < p so a+b < 2p < 2^65; subtract p once if needed
  if (c || s >= GL_P) s -= GL_P;
  return s;
}
static inline u64 gl_sub(u64 a, u64 b) { return a >= b ? a - b : a + (GL_P - b); }
static inline u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }

// 128-bit -> canonical, using 2^64 = 2^32 - 1 and 2^96 = -1 (mod p).
static inline u64 gl_reduce128(u128 x) {
  u64 lo = (u64)x, hi = (u64)(x >> 64);
  u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  u64 t0 = lo - hi_hi;
  if (lo < hi_hi) t0 -= GL_EPS;  // borrow: add p == subtract 2^32-1 (mod 2^64)
  u64 t1 = hi_lo * GL_EPS;
  u64 r = t0 + t1;
  if (r < t1) r += GL_EPS;  // carry: subtract p == add 2^32-1
  if (r >= GL_P) r -= GL_P;
  return r;
}
static inline u64 gl_mul(u64 a, u64 b) { return gl_reduce128((u128)a * b); }
static inline u64 gl_from_u64(u64 x) { return x >= GL_P ? x - GL_P : x; }

static inline u64 gl_pow(u64 b, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = gl_mul(r, b);
    b = gl_mul(b, b);
    e >>= 1;
  }
  return r;
}
static inline u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }
static inline u64 gl_exp_pow2(u64 a, unsigned k) {
  while (k--) a = gl_mul(a, a);
  return a;
}
// generator of the multiplicative subgroup of order 2^bits
static inline u64 gl_two_adic_generator(unsigned bits) { return gl_exp_pow2(GL_TWO_ADIC_GEN_32, 32 - bits); }

// ---- degree-2 extension, basis (1, X), X^2 = 7 ----
struct E2 {
  u64 c0, c1;
};
static inline E2 e2(u64 a, u64 b = 0) { return E2{a, b}; }
static inline bool e2_eq(E2 a, E2 b) { return a.c0 == b.c0 && a.c1 == b.c1; }
static inline E2 e2_add(E2 a, E2 b) { return E2{gl_add(a.c0, b.c0), gl_add(a.c1, b.c1)}; }
static inline E2 e2_sub(E2 a, E2 b) { return E2{gl_sub(a.c0, b.c0), gl_sub(a.c1, b.c1)}; }
static inline E2 e2_neg(E2 a) { return E2{gl_neg(a.c0), gl_neg(a.c1)}; }
static inline E2 e2_mul(E2 a, E2 b) {
  u64 v0 = gl_mul(a.c0, b.c0), v1 = gl_mul(a.c1, b.c1);
  u64 c0 = gl_add(v0, gl_mul(GL_EXT_W, v1));
  u64 c1 = gl_add(gl_mul(a.c0, b.c1), gl_mul(a.c1, b.c0));
  return E2{c0, c1};
}
static inline E2 e2_mul_base(E2 a, u64 b) { return E2{gl_mul(a.c0, b), gl_mul(a.c1, b)}; }
static inline E2 e2_square(E2 a) { return e2_mul(a, a); }
static inline E2 e2_inv(E2 a) {
  // 1/(a0 + a1 X) = (a0 - a1 X) / (a0^2 - 7 a1^2)
  u64 norm = gl_sub(gl_mul(a.c0, a.c0), gl_mul(GL_EXT_W, gl_mul(a.c1, a.c1)));
  u64 ni = gl_inv(norm);
  return E2{gl_mul(a.c0, ni), gl_mul(gl_neg(a.c1), ni)};
}
static inline E2 e2_pow(E2 b, u64 e) {
  E2 r = e2(1);
  while (e) {
    if (e & 1) r = e2_mul(r, b);
    b = e2_mul(b, b);
    e >>= 1;
  }
  return r;
}
static inline E2 e2_exp_pow2(E2 a, unsigned k) {
  while (k--) a = e2_mul(a, a);
  return a;
}

static inline unsigned log2_strict(size_t n) {
  unsigned l = 0;
  while ((size_t(1) << l) < n) l++;
  return l;
}
static inline size_t bitrev(size_t x, unsigned bits) {
  size_t r = 0;
  for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
  return r;
}

}  // namespace mso
