// ORACLE — test infrastructure only (CPU restatement of the reference's arithmetic).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything
// under oracle/. The product (multi-stark_amd/csrc) never includes or links this.
//
// BabyBear field p = 2^31 - 2^27 + 1 for the reference's second configuration
// (src/test_circuits/baby_bear_config.rs:28-38: Val = BabyBear, Challenge = BinomialExtensionField<Val, 4>,
// Perm = Poseidon2BabyBear<16>). Restates Plonky3 p3-baby-bear / p3-monty-31 / p3-poseidon2 0.5.1
// (git e9d75614, not vendored in /root/reference):
//   GENERATOR = 31, TWO_ADICITY = 27, two_adic_generator(27) = 0x1a427a41 (= 31^15, checked numerically),
//   BinomiallyExtendable<4>::W = 11 (recovered by the reference as X^D, src/system.rs:334-349).
// Elements are held as canonical integers in a u64 here (the oracle is not built for speed);
// the serde form of MontyField31 is the Montgomery representation x * 2^32 mod p as a u32
// [UPSTREAM-RECALL p3-monty-31 0.5.1 "It's faster to Serialize and Deserialize in monty form"] -> bb_to_wire().
// PARITY UNPINNED at the Plonky3 boundary (no golden vectors in the reference).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace mso {

typedef uint64_t u64;

static const u64 F_P = 2013265921ULL;  // 0x78000001
static const u64 F_GENERATOR = 31;
static const u64 F_TWO_ADIC_GEN_TOP = 0x1a427a41ULL;
static const unsigned F_TWO_ADICITY = 27;
static const u64 EXT_W = 11;
static const unsigned EXT_D = 4;
static const unsigned F_WIRE_BYTES = 4;

static inline u64 f_add(u64 a, u64 b) {
  u64 s = a + b;
  return s >= F_P ? s - F_P : s;
}
static inline u64 f_sub(u64 a, u64 b) { return a >= b ? a - b : a + F_P - b; }
static inline u64 f_neg(u64 a) { return a ? F_P - a : 0; }
static inline u64 f_mul(u64 a, u64 b) { return a * b % F_P; }
static inline u64 f_from_u64(u64 x) { return x % F_P; }
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
static inline u64 f_two_adic_generator(unsigned bits) { return f_exp_pow2(F_TWO_ADIC_GEN_TOP, F_TWO_ADICITY - bits); }

// MontyField31 <-> canonical (the serde form)
static inline uint32_t bb_to_wire(u64 canonical) { return (uint32_t)((canonical << 32) % F_P); }
static inline u64 bb_from_wire(uint32_t monty) {
  static const u64 R_INV = f_inv((u64(1) << 32) % F_P);
  return f_mul(monty, R_INV);
}

// ---- Poseidon2BabyBear<16> (p3-poseidon2 0.5.1 generic structure + p3-baby-bear's linear layers) ----
// S-box x^7, 8 external rounds (4 + 4), 13 internal rounds. The round constants come from
// Perm::new_from_rng_128(SmallRng::seed_from_u64(42)) in the reference (baby_bear_config.rs:54-55), a stream that
// cannot be reproduced without the rand crate: they are INPUTS here (8*16 external, then 13 internal, canonical).
struct Poseidon2Constants {
  u64 external[8][16];
  u64 internal[13];
};
// V = [-2, 1, 2, 1/2, 3, 4, -1/2, -3, -4, 1/2^8, 1/4, 1/8, 1/2^27, -1/2^8, -1/16, -1/2^27]: the internal layer is
// state <- 1 * sum(state) + diag(V) * state  [UPSTREAM-RECALL p3-baby-bear poseidon2.rs INTERNAL_DIAG_MONTY_16]
static inline const u64* bb_internal_diag16() {
  static u64 v[16];
  static bool init = false;
  if (!init) {
    u64 half = f_inv(2), i8 = f_inv(256), i27 = f_inv(u64(1) << 27);
    u64 t[16] = {f_neg(2), 1, 2, half, 3, 4, f_neg(half), f_neg(3), f_neg(4), i8, f_inv(4), f_inv(8), i27, f_neg(i8),
                 f_neg(f_inv(16)), f_neg(i27)};
    for (int i = 0; i < 16; i++) v[i] = t[i];
    init = true;
  }
  return v;
}
static inline u64 bb_sbox7(u64 x) {
  u64 x2 = f_mul(x, x), x3 = f_mul(x2, x), x4 = f_mul(x2, x2);
  return f_mul(x3, x4);
}
// MDSMat4 = circ-like [[2,3,1,1],[1,2,3,1],[1,1,2,3],[3,1,1,2]] on every 4-chunk, then add the column sums of the
// chunks to every chunk (p3-poseidon2 mds_light_permutation, width 16)
static inline void bb_mds_light16(u64* s) {
  for (int c = 0; c < 16; c += 4) {
    u64 a = s[c], b = s[c + 1], cc = s[c + 2], d = s[c + 3];
    u64 sum = f_add(f_add(a, b), f_add(cc, d));
    u64 n0 = f_add(f_add(sum, a), f_add(b, b));    // 2a + 3b + c + d
    u64 n1 = f_add(f_add(sum, b), f_add(cc, cc));  // a + 2b + 3c + d
    u64 n2 = f_add(f_add(sum, cc), f_add(d, d));   // a + b + 2c + 3d
    u64 n3 = f_add(f_add(sum, d), f_add(a, a));    // 3a + b + c + 2d
    s[c] = n0, s[c + 1] = n1, s[c + 2] = n2, s[c + 3] = n3;
  }
  u64 col[4];
  for (int k = 0; k < 4; k++) col[k] = f_add(f_add(s[k], s[4 + k]), f_add(s[8 + k], s[12 + k]));
  for (int i = 0; i < 16; i++) s[i] = f_add(s[i], col[i & 3]);
}
static inline void bb_poseidon2_permute(const Poseidon2Constants& k, u64* s) {
  const u64* V = bb_internal_diag16();
  bb_mds_light16(s);
  for (int r = 0; r < 4; r++) {
    for (int i = 0; i < 16; i++) s[i] = bb_sbox7(f_add(s[i], k.external[r][i]));
    bb_mds_light16(s);
  }
  for (int r = 0; r < 13; r++) {
    s[0] = bb_sbox7(f_add(s[0], k.internal[r]));
    u64 sum = 0;
    for (int i = 0; i < 16; i++) sum = f_add(sum, s[i]);
    for (int i = 0; i < 16; i++) s[i] = f_add(sum, f_mul(V[i], s[i]));
  }
  for (int r = 4; r < 8; r++) {
    for (int i = 0; i < 16; i++) s[i] = bb_sbox7(f_add(s[i], k.external[r][i]));
    bb_mds_light16(s);
  }
}

}  // namespace mso
