// Host-side microbenchmark: narrowing a 2^20 x 14 trace of 64-bit words to bytes on T threads in eight chunks, the chunks made
// of (a) consecutive rows, (b) runs of R rows taken from every block of 8 R rows ("row groups", prover.hip HostUpload), with and
// without software prefetch of the following run. Prints the time until the last chunk is complete.
//   g++ -O2 -std=c++17 -pthread tools/micro/pack_runs.cpp multi-stark_amd/csrc/pack_host.cpp -o pack_runs && ./pack_runs 16
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
namespace msamd {
uint64_t narrow_range(const uint64_t* in, uint8_t* out, unsigned pb, size_t n);
}
using namespace msamd;

int main(int argc, char** argv) {
  const int nt = argc > 1 ? atoi(argv[1]) : 16;
  const size_t h = size_t(1) << 20, w = 14, cnt = h * w, per_chunk = cnt / 8;
  uint64_t* in = (uint64_t*)aligned_alloc(4096, cnt * 8);
  uint8_t* out = (uint8_t*)aligned_alloc(4096, cnt);
  for (size_t i = 0; i < cnt; i++) in[i] = i & 0xff;
  memset(out, 0, cnt);
  struct Cfg {
    const char* name;
    size_t run_rows;  // 0 = consecutive rows
    bool prefetch;
  } cfgs[] = {{"consecutive rows", 0, false}, {"runs of 32 rows", 32, false}, {"runs of 32 rows + prefetch", 32, true}, {"runs of 64 rows", 64, false},
              {"runs of 128 rows", 128, false}, {"runs of 512 rows", 512, false}};
  std::atomic<int> go{0}, done{0};
  std::atomic<size_t> next{0};
  std::atomic<bool> quit{false};
  Cfg cur = cfgs[0];
  const size_t nsub = 2 * (size_t)nt;
  auto piece = [&](size_t it) {
    const size_t k = it / nsub, sub = it % nsub;
    if (!cur.run_rows) {
      const size_t b = k * per_chunk, p = per_chunk / nsub;
      narrow_range(in + b + sub * p, out + b + sub * p, 1, p);
      return;
    }
    const size_t run_words = cur.run_rows * w, run_stride = run_words * 8, runs = cnt / run_stride, per = runs / nsub;
    for (size_t m = sub * per; m < (sub + 1) * per; m++) {
#if defined(__x86_64__)
      if (cur.prefetch && m + 1 < (sub + 1) * per) {
        const char* nx = (const char*)(in + (m + 1) * run_stride + k * run_words);
        for (size_t o = 0; o < run_words * 8; o += 64) _mm_prefetch(nx + o, _MM_HINT_T0);
      }
#endif
      narrow_range(in + m * run_stride + k * run_words, out + k * per_chunk + m * run_words, 1, run_words);
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < nt; t++)
    th.emplace_back([&]() {
      int seen = 0;
      for (;;) {
        while (go.load(std::memory_order_acquire) == seen)
          if (quit.load()) return;
        seen++;
        for (size_t it; (it = next.fetch_add(1)) < 8 * nsub;) piece(it);
        done.fetch_add(1, std::memory_order_release);
      }
    });
  for (int pass = 0; pass < 2; pass++)  // (the first pass over the configurations touches every page and wakes the cores: not printed)
  for (auto& c : cfgs) {
    cur = c;
    double best = 1e30, sum = 0;
    const int reps = 12;
    for (int rep = 0; rep < reps; rep++) {
      next.store(0);
      done.store(0);
      auto t0 = std::chrono::steady_clock::now();
      go.fetch_add(1, std::memory_order_release);
      while (done.load(std::memory_order_acquire) != nt) {
      }
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (rep >= 2) sum += us;
      if (us < best) best = us;
    }
    if (pass) printf("%-30s best %7.1f us  mean %7.1f us  (%.0f GB/s of source at best)\n", c.name, best, sum / (reps - 2), cnt * 8 / best / 1e3);
  }
  quit.store(true);
  go.fetch_add(1);
  for (auto& x : th) x.join();
  return 0;
}
