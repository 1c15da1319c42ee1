// Launch overhead of short dependent kernels on one stream: plain launches against a captured hipGraph replay.
// hipcc -O3 --offload-arch=gfx950 tools/micro/graph_launch.hip -o tools/micro/graph_launch && tools/micro/graph_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(unsigned long long* p, int k) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += k; }
__global__ void medium(unsigned long long* p, int n) {  // ~10 us of dependent work in one workgroup
  unsigned long long v = p[threadIdx.x & 7];
  for (int i = 0; i < n; i++) v = v * 6364136223846793005ULL + 1442695040888963407ULL;
  if (v == 12345) p[1] = v;
}
int main() {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  unsigned long long* d;
  CK(hipMalloc(&d, 64));
  CK(hipMemset(d, 0, 64));
  const int N = 60, REP = 200;
  for (int which = 0; which < 2; which++) {
    auto body = [&](hipStream_t st) {
      for (int i = 0; i < N; i++) {
        if (which == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, d, i);
        else hipLaunchKernelGGL(medium, dim3(1), dim3(256), 0, st, d, 2000);
      }
    };
    body(s);
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REP; r++) body(s);
    CK(hipStreamSynchronize(s));
    double plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (REP * N);
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    body(s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < REP; r++) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    double graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (REP * N);
    printf("%s kernels, %d dependent launches: %.2f us each as plain launches, %.2f us each inside a hipGraph\n",
           which == 0 ? "trivial" : "~10 us", N, plain, graph);
  }
  return 0;
}
