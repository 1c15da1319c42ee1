// ORACLE — test infrastructure only. Plain BLAKE3 (unkeyed hash mode, 32-byte output).
// Restates the published BLAKE3 algorithm (blake3 crate 1.8.5 is the reference's dependency,
// Cargo.lock:71-72; not vendored). The round function is pinned against the reference's own
// literal KATs (src/test_circuits/blake3.rs:2616-2644 g function, :2646-2746 compression) and the
// in-tree reference hasher src/test_circuits/blake3.rs:32-351 served as the readable spec.
// Used by the reference as: SerializingHasher<Blake3> leaf hash, CompressionFunctionFromHasher<Blake3,2,32>
// (src/types.rs:82-83,199-207) and HashChallenger<u8,Blake3,32> (src/types.rs:28-29).
#pragma once
#include <cstdint>
#include <cstring>
#include <cstddef>

namespace mso {

static const uint32_t B3_IV[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au,
                                  0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u};
static const int B3_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
enum { B3_CHUNK_START = 1, B3_CHUNK_END = 2, B3_PARENT = 4, B3_ROOT = 8 };

static inline uint32_t b3_rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

static inline void b3_g(uint32_t* v, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
  v[a] = v[a] + v[b] + mx;
  v[d] = b3_rotr(v[d] ^ v[a], 16);
  v[c] = v[c] + v[d];
  v[b] = b3_rotr(v[b] ^ v[c], 12);
  v[a] = v[a] + v[b] + my;
  v[d] = b3_rotr(v[d] ^ v[a], 8);
  v[c] = v[c] + v[d];
  v[b] = b3_rotr(v[b] ^ v[c], 7);
}

// 7 rounds over a 16-word state v with 16 message words m (m is permuted between rounds).
static inline void b3_rounds(uint32_t v[16], const uint32_t m_in[16]) {
  uint32_t m[16];
  memcpy(m, m_in, sizeof(m));
  for (int r = 0; r < 7; r++) {
    b3_g(v, 0, 4, 8, 12, m[0], m[1]);
    b3_g(v, 1, 5, 9, 13, m[2], m[3]);
    b3_g(v, 2, 6, 10, 14, m[4], m[5]);
    b3_g(v, 3, 7, 11, 15, m[6], m[7]);
    b3_g(v, 0, 5, 10, 15, m[8], m[9]);
    b3_g(v, 1, 6, 11, 12, m[10], m[11]);
    b3_g(v, 2, 7, 8, 13, m[12], m[13]);
    b3_g(v, 3, 4, 9, 14, m[14], m[15]);
    if (r < 6) {
      uint32_t t[16];
      for (int i = 0; i < 16; i++) t[i] = m[B3_PERM[i]];
      memcpy(m, t, sizeof(m));
    }
  }
}

// Full compression; writes the 16-word output (first 8 words = new chaining value).
static inline void b3_compress(const uint32_t cv[8], const uint32_t block[16], uint64_t counter,
                               uint32_t block_len, uint32_t flags, uint32_t out[16]) {
  uint32_t v[16];
  for (int i = 0; i < 8; i++) v[i] = cv[i];
  v[8] = B3_IV[0];
  v[9] = B3_IV[1];
  v[10] = B3_IV[2];
  v[11] = B3_IV[3];
  v[12] = (uint32_t)counter;
  v[13] = (uint32_t)(counter >> 32);
  v[14] = block_len;
  v[15] = flags;
  b3_rounds(v, block);
  for (int i = 0; i < 8; i++) {
    out[i] = v[i] ^ v[i + 8];
    out[i + 8] = v[i + 8] ^ cv[i];
  }
}

static inline void b3_load_block(const uint8_t* p, size_t len, uint32_t w[16]) {
  uint8_t buf[64];
  memset(buf, 0, 64);
  if (len) memcpy(buf, p, len);
  for (int i = 0; i < 16; i++)
    w[i] = (uint32_t)buf[4 * i] | ((uint32_t)buf[4 * i + 1] << 8) | ((uint32_t)buf[4 * i + 2] << 16) |
           ((uint32_t)buf[4 * i + 3] << 24);
}

// chaining value of one chunk (len <= 1024); extra_flags carries ROOT when the chunk is the whole input
static inline void b3_chunk_cv(const uint8_t* in, size_t len, uint64_t chunk_index, uint32_t root_flag,
                               uint32_t out[8]) {
  uint32_t cv[8];
  memcpy(cv, B3_IV, sizeof(cv));
  size_t nblocks = len == 0 ? 1 : (len + 63) / 64;
  for (size_t b = 0; b < nblocks; b++) {
    size_t off = b * 64;
    size_t bl = len - off < 64 ? len - off : 64;
    uint32_t w[16], o[16];
    b3_load_block(in + off, bl, w);
    uint32_t flags = 0;
    if (b == 0) flags |= B3_CHUNK_START;
    if (b == nblocks - 1) flags |= B3_CHUNK_END | root_flag;
    b3_compress(cv, w, chunk_index, (uint32_t)bl, flags, o);
    memcpy(cv, o, sizeof(cv));
  }
  memcpy(out, cv, sizeof(cv));
}

static inline void b3_subtree_cv(const uint8_t* in, size_t len, uint64_t chunk_index, uint32_t root_flag,
                                 uint32_t out[8]) {
  if (len <= 1024) {
    b3_chunk_cv(in, len, chunk_index, root_flag, out);
    return;
  }
  // left subtree: largest power-of-two number of full chunks strictly less than the total
  size_t full = (len - 1) / 1024;
  size_t lc = 1;
  while (lc * 2 <= full) lc *= 2;
  size_t left_len = lc * 1024;
  uint32_t block[16], o[16];
  b3_subtree_cv(in, left_len, chunk_index, 0, block);
  b3_subtree_cv(in + left_len, len - left_len, chunk_index + lc, 0, block + 8);
  b3_compress(B3_IV, block, 0, 64, B3_PARENT | root_flag, o);
  memcpy(out, o, 32);
}

static inline void blake3_hash(const uint8_t* in, size_t len, uint8_t out[32]) {
  uint32_t cv[8];
  b3_subtree_cv(in, len, 0, B3_ROOT, cv);
  for (int i = 0; i < 8; i++) {
    out[4 * i] = (uint8_t)cv[i];
    out[4 * i + 1] = (uint8_t)(cv[i] >> 8);
    out[4 * i + 2] = (uint8_t)(cv[i] >> 16);
    out[4 * i + 3] = (uint8_t)(cv[i] >> 24);
  }
}

}  // namespace mso
