// BabyBear (p = 2^31 - 2^27 + 1) arithmetic for the reference's second configuration
// (/root/reference/src/test_circuits/baby_bear_config.rs:28-38): Val = BabyBear, Challenge = BinomialExtensionField<Val, 4>
// (X^4 = 11), Perm = Poseidon2BabyBear<16>. Values live on the device in Montgomery form (x * 2^32 mod p) - the form
// p3-monty-31 itself keeps and serialises - so digests and opened values go into the proof as the raw device words.
// Host and device share this header (the host runs the DuplexChallenger transcript).
#pragma once
#if defined(__HIPCC_RTC__)  // hiprtc (the per-circuit quotient kernel, quotient_jit.hip) has no standard library headers
typedef unsigned long uint64_t;
typedef unsigned int uint32_t;
typedef unsigned long size_t;
#else
#include <hip/hip_runtime.h>

#include <cstdint>
#endif

namespace msbb {

typedef uint32_t u32;
typedef uint64_t u64;

#define BB_HD __host__ __device__ __forceinline__

static constexpr u32 BB_P = 0x78000001u;
static constexpr u32 BB_PINV = 0x88000001u;    // p^-1 mod 2^32
static constexpr u32 BB_R1 = 0x0ffffffeu;      // 2^32 mod p  (Montgomery form of 1)
static constexpr u32 BB_R2 = 0x45dddde3u;      // 2^64 mod p  (to_monty(x) = mont_mul(x, R2))
static constexpr unsigned BB_TWO_ADICITY = 27;
static constexpr u32 BB_GENERATOR = 31;        // canonical
static constexpr u32 BB_TWO_ADIC_GEN = 0x1a427a41u;  // canonical, order 2^27 (= 31^15)
static constexpr u32 BB_EXT_W = 11;            // canonical; X^4 = 11

// one conditional correction as an unsigned minimum: the wrong candidate wraps around to a value above 2^31
BB_HD u32 bb_add(u32 a, u32 b) {
  u32 s = a + b, t = s - BB_P;
  return t < s ? t : s;
}
BB_HD u32 bb_sub(u32 a, u32 b) {
  u32 d = a - b, t = d + BB_P;
  return t < d ? t : d;
}
BB_HD u32 bb_neg(u32 a) { return a ? BB_P - a : 0; }
// Montgomery reduction of x < p * 2^32: x / 2^32 mod p
BB_HD u32 bb_mred(u64 x) {
  u32 lo = (u32)x, hi = (u32)(x >> 32);
  u32 t = lo * BB_PINV;
  u32 uh = (u32)(((u64)t * BB_P) >> 32);  // the low word of t * p equals lo by construction: only the high words differ
  u32 r = hi - uh, r2 = r + BB_P;  // hi < uh: r wrapped and r + p is the answer; else r + p > r
  return r2 < r ? r2 : r;
}
BB_HD u32 bb_mul(u32 a, u32 b) { return bb_mred((u64)a * b); }
BB_HD u32 bb_to_monty(u32 canonical) { return bb_mul(canonical, BB_R2); }
BB_HD u32 bb_from_monty(u32 m) { return bb_mred((u64)m); }
BB_HD u32 bb_pow(u32 b, u64 e) {
  u32 r = BB_R1;
  while (e) {
    if (e & 1) r = bb_mul(r, b);
    b = bb_mul(b, b);
    e >>= 1;
  }
  return r;
}
BB_HD u32 bb_inv(u32 a) { return bb_pow(a, BB_P - 2); }
BB_HD u32 bb_exp_pow2(u32 a, unsigned k) {
  while (k--) a = bb_mul(a, a);
  return a;
}
// Montgomery form of the generator of the order-2^bits subgroup
BB_HD u32 bb_two_adic_generator(unsigned bits) { return bb_exp_pow2(bb_to_monty(BB_TWO_ADIC_GEN), BB_TWO_ADICITY - bits); }

// ---- degree-4 binomial extension, basis (1, X, X^2, X^3), X^4 = 11; coordinates in Montgomery form
struct E4 {
  u32 c[4];
};
BB_HD E4 e4_zero() { return E4{{0, 0, 0, 0}}; }
BB_HD E4 e4_base(u32 m) { return E4{{m, 0, 0, 0}}; }
BB_HD E4 e4_one() { return e4_base(BB_R1); }
BB_HD bool e4_eq(E4 a, E4 b) { return a.c[0] == b.c[0] && a.c[1] == b.c[1] && a.c[2] == b.c[2] && a.c[3] == b.c[3]; }
BB_HD E4 e4_add(E4 a, E4 b) { return E4{{bb_add(a.c[0], b.c[0]), bb_add(a.c[1], b.c[1]), bb_add(a.c[2], b.c[2]), bb_add(a.c[3], b.c[3])}}; }
BB_HD E4 e4_sub(E4 a, E4 b) { return E4{{bb_sub(a.c[0], b.c[0]), bb_sub(a.c[1], b.c[1]), bb_sub(a.c[2], b.c[2]), bb_sub(a.c[3], b.c[3])}}; }
BB_HD E4 e4_neg(E4 a) { return E4{{bb_neg(a.c[0]), bb_neg(a.c[1]), bb_neg(a.c[2]), bb_neg(a.c[3])}}; }
BB_HD E4 e4_mul_base(E4 a, u32 b) { return E4{{bb_mul(a.c[0], b), bb_mul(a.c[1], b), bb_mul(a.c[2], b), bb_mul(a.c[3], b)}}; }
// sums of up to four products stay below p * 2^32 only pairwise; accumulate reduced values
BB_HD u32 bb_mul11(u32 a) {  // 11 a = 8a + 2a + a
  u32 a2 = bb_add(a, a), a4 = bb_add(a2, a2), a8 = bb_add(a4, a4);
  return bb_add(bb_add(a8, a2), a);
}
BB_HD E4 e4_mul(E4 a, E4 b) {
  // each 64-bit product is < p^2 < 2^62, so two of them can be added before one reduction
  u64 p00 = (u64)a.c[0] * b.c[0], p01 = (u64)a.c[0] * b.c[1], p02 = (u64)a.c[0] * b.c[2], p03 = (u64)a.c[0] * b.c[3];
  u64 p10 = (u64)a.c[1] * b.c[0], p11 = (u64)a.c[1] * b.c[1], p12 = (u64)a.c[1] * b.c[2], p13 = (u64)a.c[1] * b.c[3];
  u64 p20 = (u64)a.c[2] * b.c[0], p21 = (u64)a.c[2] * b.c[1], p22 = (u64)a.c[2] * b.c[2], p23 = (u64)a.c[2] * b.c[3];
  u64 p30 = (u64)a.c[3] * b.c[0], p31 = (u64)a.c[3] * b.c[1], p32 = (u64)a.c[3] * b.c[2], p33 = (u64)a.c[3] * b.c[3];
  // low parts (degree k), high parts (degree k + 4, multiplied by 11)
  u32 h0 = bb_add(bb_mred(p13 + p31), bb_mred(p22));
  u32 h1 = bb_mred(p23 + p32);
  u32 h2 = bb_mred(p33);
  E4 r;
  r.c[0] = bb_add(bb_mred(p00), bb_mul11(h0));
  r.c[1] = bb_add(bb_mred(p01 + p10), bb_mul11(h1));
  r.c[2] = bb_add(bb_add(bb_mred(p02 + p20), bb_mred(p11)), bb_mul11(h2));
  r.c[3] = bb_add(bb_mred(p03 + p30), bb_mred(p12 + p21));
  return r;
}
BB_HD E4 e4_square(E4 a) { return e4_mul(a, a); }
// tower inverse: Y = X^2, a = A0(Y) + X A1(Y); 1/a = (A0 - X A1) / (A0^2 - Y A1^2), the denominator in F[Y]/(Y^2 - 11)
BB_HD E4 e4_inv(E4 a) {
  u32 a0 = a.c[0], a1 = a.c[1], a2 = a.c[2], a3 = a.c[3];
  u32 s0 = bb_add(bb_mul(a0, a0), bb_mul11(bb_mul(a2, a2))), s1 = bb_mul(a0, a2);
  s1 = bb_add(s1, s1);
  u32 t0 = bb_add(bb_mul(a1, a1), bb_mul11(bb_mul(a3, a3))), t1 = bb_mul(a1, a3);
  t1 = bb_add(t1, t1);
  u32 d0 = bb_sub(s0, bb_mul11(t1)), d1 = bb_sub(s1, t0);
  u32 ni = bb_inv(bb_sub(bb_mul(d0, d0), bb_mul11(bb_mul(d1, d1))));
  u32 i0 = bb_mul(d0, ni), i1 = bb_mul(bb_neg(d1), ni);
  E4 r;
  r.c[0] = bb_add(bb_mul(a0, i0), bb_mul11(bb_mul(a2, i1)));
  r.c[2] = bb_add(bb_mul(a0, i1), bb_mul(a2, i0));
  r.c[1] = bb_neg(bb_add(bb_mul(a1, i0), bb_mul11(bb_mul(a3, i1))));
  r.c[3] = bb_neg(bb_add(bb_mul(a1, i1), bb_mul(a3, i0)));
  return r;
}
BB_HD E4 e4_exp_pow2(E4 a, unsigned k) {
  while (k--) a = e4_mul(a, a);
  return a;
}
BB_HD E4 e4_pow(E4 b, u64 e) {
  E4 r = e4_one();
  while (e) {
    if (e & 1) r = e4_mul(r, b);
    b = e4_mul(b, b);
    e >>= 1;
  }
  return r;
}

// ---- Poseidon2BabyBear<16>: x^7, 4 + 13 + 4 rounds. Round constants are inputs of the configuration (the reference
// draws them from an RNG, baby_bear_config.rs:54-55); everything here is in Montgomery form.
struct Poseidon2 {
  u32 external[8][16];
  u32 internal[13];
  u32 diag[16];  // V = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 1/2^8, 1/4, 1/8, 1/2^27, -1/2^8, -1/16, -1/2^27]
};
// x / 2 and x / 2^k without a general multiplication: a shift into the Montgomery reduction
BB_HD u32 bb_halve(u32 x) { return (x & 1) ? (x + BB_P) >> 1 : x >> 1; }
template <int K>
BB_HD u32 bb_div2k(u32 x) {
  return bb_mred((u64)x << (32 - K));
}
BB_HD u32 bb_sbox7(u32 x) {
  u32 x2 = bb_mul(x, x), x3 = bb_mul(x2, x), x4 = bb_mul(x2, x2);
  return bb_mul(x3, x4);
}
// M4 = [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] on each 4-chunk, then every chunk += the column sums
BB_HD void bb_mds_light16(u32* s) {
#pragma unroll
  for (int c = 0; c < 16; c += 4) {  // 9 additions + 2 doublings per chunk
    u32 a = s[c], b = s[c + 1], cc = s[c + 2], d = s[c + 3];
    u32 t01 = bb_add(a, b), t23 = bb_add(cc, d), t0123 = bb_add(t01, t23);
    u32 t01123 = bb_add(t0123, b), t01233 = bb_add(t0123, d);
    s[c + 3] = bb_add(t01233, bb_add(a, a));   // 3a + b + c + 2d
    s[c + 1] = bb_add(t01123, bb_add(cc, cc)); // a + 2b + 3c + d
    s[c] = bb_add(t01123, t01);                // 2a + 3b + c + d
    s[c + 2] = bb_add(t01233, t23);            // a + b + 2c + 3d
  }
  u32 col[4];
#pragma unroll
  for (int k = 0; k < 4; k++) col[k] = bb_add(bb_add(s[k], s[4 + k]), bb_add(s[8 + k], s[12 + k]));
#pragma unroll
  for (int i = 0; i < 16; i++) s[i] = bb_add(s[i], col[i & 3]);
}
BB_HD void bb_poseidon2(const Poseidon2& k, u32* s) {
  bb_mds_light16(s);
  for (int r = 0; r < 4; r++) {
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = bb_sbox7(bb_add(s[i], k.external[r][i]));
    bb_mds_light16(s);
  }
  for (int r = 0; r < 13; r++) {
    s[0] = bb_sbox7(bb_add(s[0], k.internal[r]));
    u32 sum = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) sum = bb_add(sum, s[i]);
    // state <- sum + diag(V) state, V = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 1/2^8, 1/4, 1/8, 1/2^27, -1/2^8, -1/16, -1/2^27]:
    // small multiples are additions, powers of 1/2 are shifts into the Montgomery reduction (k.diag holds the same
    // constants for the lane-parallel form)
    u32 d2, d4;
    d2 = bb_add(s[0], s[0]), s[0] = bb_sub(sum, d2);
    s[1] = bb_add(sum, s[1]);
    s[2] = bb_add(sum, bb_add(s[2], s[2]));
    s[3] = bb_add(sum, bb_halve(s[3]));
    d2 = bb_add(s[4], s[4]), s[4] = bb_add(sum, bb_add(d2, s[4]));
    d2 = bb_add(s[5], s[5]), s[5] = bb_add(sum, bb_add(d2, d2));
    s[6] = bb_sub(sum, bb_halve(s[6]));
    d2 = bb_add(s[7], s[7]), s[7] = bb_sub(sum, bb_add(d2, s[7]));
    d2 = bb_add(s[8], s[8]), d4 = bb_add(d2, d2), s[8] = bb_sub(sum, d4);
    s[9] = bb_add(sum, bb_div2k<8>(s[9]));
    s[10] = bb_add(sum, bb_halve(bb_halve(s[10])));
    s[11] = bb_add(sum, bb_div2k<3>(s[11]));
    s[12] = bb_add(sum, bb_div2k<27>(s[12]));
    s[13] = bb_sub(sum, bb_div2k<8>(s[13]));
    s[14] = bb_sub(sum, bb_div2k<4>(s[14]));
    s[15] = bb_sub(sum, bb_div2k<27>(s[15]));
  }
  for (int r = 4; r < 8; r++) {
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = bb_sbox7(bb_add(s[i], k.external[r][i]));
    bb_mds_light16(s);
  }
}

struct Digest8 {
  u32 w[8];
};

}  // namespace msbb
