// ORACLE — test infrastructure only. Binomial extension F[X]/(X^D - W) over the configured base field
// (p3-field BinomialExtensionField): D = 2, W = 7 for Goldilocks (src/types.rs:26-27); D = 4, W = 11 for BabyBear
// (src/test_circuits/baby_bear_config.rs:35). Basis (1, X, .., X^(D-1)); the coordinates of every result are fixed
// by mathematics, so the schoolbook formulas below give the same canonical values as Plonky3's specialised ones.
#pragma once

namespace mso {

struct EF {
  u64 c[EXT_D];
};
static inline EF ef(u64 a) {
  EF r;
  r.c[0] = a;
  for (unsigned k = 1; k < EXT_D; k++) r.c[k] = 0;
  return r;
}
static inline EF ef_from(const u64* p) {
  EF r;
  for (unsigned k = 0; k < EXT_D; k++) r.c[k] = p[k];
  return r;
}
static inline void ef_to(EF a, u64* p) {
  for (unsigned k = 0; k < EXT_D; k++) p[k] = a.c[k];
}
static inline EF ef_basis(unsigned k) {  // X^k
  EF r = ef(0);
  r.c[k] = 1;
  return r;
}
static inline bool ef_eq(EF a, EF b) {
  for (unsigned k = 0; k < EXT_D; k++)
    if (a.c[k] != b.c[k]) return false;
  return true;
}
static inline bool ef_is_zero(EF a) { return ef_eq(a, ef(0)); }
static inline bool ef_less(EF a, EF b) {  // any strict total order (used for ordered maps only)
  for (unsigned k = 0; k < EXT_D; k++)
    if (a.c[k] != b.c[k]) return a.c[k] < b.c[k];
  return false;
}
static inline EF ef_add(EF a, EF b) {
  for (unsigned k = 0; k < EXT_D; k++) a.c[k] = f_add(a.c[k], b.c[k]);
  return a;
}
static inline EF ef_sub(EF a, EF b) {
  for (unsigned k = 0; k < EXT_D; k++) a.c[k] = f_sub(a.c[k], b.c[k]);
  return a;
}
static inline EF ef_neg(EF a) {
  for (unsigned k = 0; k < EXT_D; k++) a.c[k] = f_neg(a.c[k]);
  return a;
}
static inline EF ef_mul(EF a, EF b) {
  u64 lo[EXT_D], hi[EXT_D];
  for (unsigned k = 0; k < EXT_D; k++) lo[k] = hi[k] = 0;
  for (unsigned i = 0; i < EXT_D; i++)
    for (unsigned j = 0; j < EXT_D; j++) {
      u64 p = f_mul(a.c[i], b.c[j]);
      if (i + j < EXT_D)
        lo[i + j] = f_add(lo[i + j], p);
      else
        hi[i + j - EXT_D] = f_add(hi[i + j - EXT_D], p);
    }
  EF r;
  for (unsigned k = 0; k < EXT_D; k++) r.c[k] = f_add(lo[k], f_mul(EXT_W, hi[k]));
  return r;
}
static inline EF ef_mul_base(EF a, u64 b) {
  for (unsigned k = 0; k < EXT_D; k++) a.c[k] = f_mul(a.c[k], b);
  return a;
}
static inline EF ef_square(EF a) { return ef_mul(a, a); }
static inline EF ef_inv(EF a) {
  EF r;
  if (EXT_D == 2) {
    // 1/(a0 + a1 X) = (a0 - a1 X) / (a0^2 - W a1^2)
    u64 ni = f_inv(f_sub(f_mul(a.c[0], a.c[0]), f_mul(EXT_W, f_mul(a.c[1], a.c[1]))));
    r.c[0] = f_mul(a.c[0], ni);
    r.c[1] = f_mul(f_neg(a.c[1]), ni);
  } else {
    // tower: Y = X^2, a = A0(Y) + X A1(Y) with A0 = a0 + a2 Y, A1 = a1 + a3 Y in F[Y]/(Y^2 - W);
    // 1/a = (A0 - X A1) / (A0^2 - Y A1^2)
    const unsigned e = EXT_D - 2, o = EXT_D - 1;  // = 2, 3 (spelled so that the D = 2 build stays in bounds)
    u64 a0 = a.c[0], a1 = a.c[1], a2 = a.c[e], a3 = a.c[o];
    // A0^2 = (a0^2 + W a2^2) + 2 a0 a2 Y ; A1^2 = (a1^2 + W a3^2) + 2 a1 a3 Y ; Y*A1^2 = W*2a1a3 + (a1^2 + W a3^2) Y
    u64 s0 = f_add(f_mul(a0, a0), f_mul(EXT_W, f_mul(a2, a2))), s1 = f_mul(2, f_mul(a0, a2));
    u64 t0 = f_add(f_mul(a1, a1), f_mul(EXT_W, f_mul(a3, a3))), t1 = f_mul(2, f_mul(a1, a3));
    u64 d0 = f_sub(s0, f_mul(EXT_W, t1)), d1 = f_sub(s1, t0);
    // 1/(d0 + d1 Y) = (d0 - d1 Y)/(d0^2 - W d1^2)
    u64 ni = f_inv(f_sub(f_mul(d0, d0), f_mul(EXT_W, f_mul(d1, d1))));
    u64 i0 = f_mul(d0, ni), i1 = f_mul(f_neg(d1), ni);
    // (A0 - X A1) * (i0 + i1 Y): even part A0*(i0 + i1 Y), odd part -A1*(i0 + i1 Y)
    r.c[0] = f_add(f_mul(a0, i0), f_mul(EXT_W, f_mul(a2, i1)));
    r.c[e] = f_add(f_mul(a0, i1), f_mul(a2, i0));
    r.c[1] = f_neg(f_add(f_mul(a1, i0), f_mul(EXT_W, f_mul(a3, i1))));
    r.c[o] = f_neg(f_add(f_mul(a1, i1), f_mul(a3, i0)));
  }
  return r;
}
static inline EF ef_pow(EF b, u64 e) {
  EF r = ef(1);
  while (e) {
    if (e & 1) r = ef_mul(r, b);
    b = ef_mul(b, b);
    e >>= 1;
  }
  return r;
}
static inline EF ef_exp_pow2(EF a, unsigned k) {
  while (k--) a = ef_mul(a, a);
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
