// Host-side narrowing of a row-major trace before its upload (prover.hip, HostUpload): 64-bit words whose values fit
// 1 / 2 / 4 bytes are packed to that width, and the OR of everything read is returned so that the caller can check the
// range. Plain host C++ (no device code): an AVX-512 body when the CPU has it (one truncating move per 64 bytes), the
// portable loop otherwise.
#include <cstddef>
#include <cstdint>
#include <cstring>

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace msamd {

typedef uint64_t u64;

static u64 narrow_portable(const u64* in, uint8_t* out, unsigned pb, size_t n) {
  u64 acc = 0;
  if (pb == 1) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
      const u64 a0 = in[i], a1 = in[i + 1], a2 = in[i + 2], a3 = in[i + 3], a4 = in[i + 4], a5 = in[i + 5], a6 = in[i + 6], a7 = in[i + 7];
      acc |= a0 | a1 | a2 | a3 | a4 | a5 | a6 | a7;
      const u64 w = (a0 & 0xff) | (a1 & 0xff) << 8 | (a2 & 0xff) << 16 | (a3 & 0xff) << 24 | (a4 & 0xff) << 32 | (a5 & 0xff) << 40 |
                    (a6 & 0xff) << 48 | (a7 & 0xff) << 56;
      memcpy(out + i, &w, 8);
    }
    for (; i < n; i++) {
      acc |= in[i];
      out[i] = (uint8_t)in[i];
    }
  } else if (pb == 2) {
    uint16_t* o = reinterpret_cast<uint16_t*>(out);
    for (size_t i = 0; i < n; i++) {
      acc |= in[i];
      o[i] = (uint16_t)in[i];
    }
  } else {
    uint32_t* o = reinterpret_cast<uint32_t*>(out);
    for (size_t i = 0; i < n; i++) {
      acc |= in[i];
      o[i] = (uint32_t)in[i];
    }
  }
  return acc;
}

#if defined(__x86_64__)
__attribute__((target("avx512f,avx512bw,avx512vl"))) static u64 narrow_avx512(const u64* in, uint8_t* out, unsigned pb, size_t n) {
  __m512i acc0 = _mm512_setzero_si512(), acc1 = _mm512_setzero_si512();
  size_t i = 0;
  if (pb == 1) {
    for (; i + 16 <= n; i += 16) {
      const __m512i a = _mm512_loadu_si512(in + i), b = _mm512_loadu_si512(in + i + 8);
      acc0 = _mm512_or_si512(acc0, a);
      acc1 = _mm512_or_si512(acc1, b);
      _mm_storel_epi64(reinterpret_cast<__m128i*>(out + i), _mm512_cvtepi64_epi8(a));
      _mm_storel_epi64(reinterpret_cast<__m128i*>(out + i + 8), _mm512_cvtepi64_epi8(b));
    }
  } else if (pb == 2) {
    for (; i + 16 <= n; i += 16) {
      const __m512i a = _mm512_loadu_si512(in + i), b = _mm512_loadu_si512(in + i + 8);
      acc0 = _mm512_or_si512(acc0, a);
      acc1 = _mm512_or_si512(acc1, b);
      _mm_storeu_si128(reinterpret_cast<__m128i*>(out + 2 * i), _mm512_cvtepi64_epi16(a));
      _mm_storeu_si128(reinterpret_cast<__m128i*>(out + 2 * i + 16), _mm512_cvtepi64_epi16(b));
    }
  } else {
    for (; i + 16 <= n; i += 16) {
      const __m512i a = _mm512_loadu_si512(in + i), b = _mm512_loadu_si512(in + i + 8);
      acc0 = _mm512_or_si512(acc0, a);
      acc1 = _mm512_or_si512(acc1, b);
      _mm256_storeu_si256(reinterpret_cast<__m256i*>(out + 4 * i), _mm512_cvtepi64_epi32(a));
      _mm256_storeu_si256(reinterpret_cast<__m256i*>(out + 4 * i + 32), _mm512_cvtepi64_epi32(b));
    }
  }
  u64 acc = _mm512_reduce_or_epi64(_mm512_or_si512(acc0, acc1));
  if (i < n) acc |= narrow_portable(in + i, out + i * pb, pb, n - i);
  return acc;
}
#endif

// in[0 .. n) -> `pb`-byte little-endian words (pb = 1, 2, 4); returns the OR of all values read
u64 narrow_range(const u64* in, uint8_t* out, unsigned pb, size_t n) {
#if defined(__x86_64__)
  static const bool wide = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl");
  if (wide) return narrow_avx512(in, out, pb, n);
#endif
  return narrow_portable(in, out, pb, n);
}

}  // namespace msamd
