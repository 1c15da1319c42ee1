// Microbenchmark (host): narrowing a row-major u64 trace to bytes with a range check, N threads, as the host-resident
// witness path would do before the upload. Reports GB/s of u64 input consumed.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static inline uint64_t pack_range(const uint64_t* in, uint8_t* out, size_t n) {
  uint64_t acc = 0;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t a0 = in[i], a1 = in[i + 1], a2 = in[i + 2], a3 = in[i + 3], a4 = in[i + 4], a5 = in[i + 5], a6 = in[i + 6], a7 = in[i + 7];
    acc |= a0 | a1 | a2 | a3 | a4 | a5 | a6 | a7;
    uint64_t w = (a0 & 0xff) | (a1 & 0xff) << 8 | (a2 & 0xff) << 16 | (a3 & 0xff) << 24 | (a4 & 0xff) << 32 | (a5 & 0xff) << 40 |
                 (a6 & 0xff) << 48 | (a7 & 0xff) << 56;
    memcpy(out + i, &w, 8);
  }
  for (; i < n; i++) {
    acc |= in[i];
    out[i] = (uint8_t)in[i];
  }
  return acc;
}

int main(int argc, char** argv) {
  const size_t n = (size_t(1) << 20) * 14;
  uint64_t* in = (uint64_t*)aligned_alloc(4096, n * 8);
  uint8_t* out = (uint8_t*)aligned_alloc(4096, n);
  for (size_t i = 0; i < n; i++) in[i] = (i * 2654435761u) & 0xff;
  memset(out, 0, n);
  printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
  for (int nt : {1, 2, 4, 8, 16, 32, 64}) {
    double best = 1e9;
    for (int rep = 0; rep < 5; rep++) {
      std::atomic<uint64_t> bad{0};
      auto t0 = std::chrono::steady_clock::now();
      std::vector<std::thread> th;
      for (int t = 0; t < nt; t++)
        th.emplace_back([&, t]() {
          size_t a = n * t / nt, b = n * (t + 1) / nt;
          a &= ~size_t(7);
          b = t + 1 == nt ? n : (b & ~size_t(7));
          bad |= pack_range(in + a, out + a, b - a) >> 8;
        });
      for (auto& x : th) x.join();
      double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (ms < best) best = ms;
      if (bad) printf("bad\n");
    }
    printf("%2d threads: %.3f ms  (%.1f GB/s of input, thread creation included)\n", nt, best, n * 8 / best / 1e6);
  }
  return 0;
}
