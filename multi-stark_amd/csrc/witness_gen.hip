// The bench workload's witness generated in HBM: build_witness + build_claims of
// /root/reference/benches/multi_stark.rs:171-238 for the system [ByteTable, U32Add]. Row i of the adder is
// (x bytes, y bytes, z = x + y mod 2^32 bytes, carry, 1) with x, y the (i+1)-th states of two xorshift32 streams; the
// byte table's column counts the 12 bytes of every row; claim i = [1, x, y, z]. xorshift32 is linear over GF(2), so a
// thread jumps to the state before its first row with the precomputed powers T^(2^j) of the step matrix and then walks
// its rows - no H2D of the 117 MB trace or the 33 MB of claims for synthetic runs.
#include "host.h"

namespace msamd {

namespace {

struct XsJump {
  u32 m[32][32];  // m[j][c] = T^(2^j) e_c
};

__host__ __device__ inline u32 xs_step(u32 a) {
  a ^= a << 13;
  a ^= a >> 17;
  a ^= a << 5;
  return a;
}
__device__ inline u32 xs_apply(const u32* __restrict__ col, u32 v) {
  u32 r = 0;
#pragma unroll
  for (int c = 0; c < 32; c++) r ^= (v >> c) & 1u ? col[c] : 0u;
  return r;
}

constexpr int GEN_ROWS = 16;  // consecutive rows per thread
__global__ __launch_bounds__(256) void u32_add_bench_k(const XsJump* __restrict__ jump, u32 a0, u32 b0, size_t num_adds, size_t height,
                                                       u64* __restrict__ add /* height x 14 */, unsigned long long* __restrict__ byte /* 256 */,
                                                       u64* __restrict__ claims /* num_adds x 4 */) {
  __shared__ unsigned int hist[256];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const size_t first = (blockIdx.x * size_t(blockDim.x) + threadIdx.x) * GEN_ROWS;
  u32 a = a0, b = b0;
  for (int j = 0; j < 32; j++)
    if ((first >> j) & 1) {
      a = xs_apply(jump->m[j], a);
      b = xs_apply(jump->m[j], b);
    }
  for (int k = 0; k < GEN_ROWS; k++) {
    const size_t i = first + k;
    if (i >= height) break;
    u64* row = add + i * 14;
    if (i < num_adds) {
      a = xs_step(a);
      b = xs_step(b);
      const u64 s = (u64)a + (u64)b;
      const u32 z = (u32)s;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const u32 xb = (a >> (8 * q)) & 0xff, yb = (b >> (8 * q)) & 0xff, zb = (z >> (8 * q)) & 0xff;
        row[q] = xb;
        row[4 + q] = yb;
        row[8 + q] = zb;
        atomicAdd(&hist[xb], 1u);
        atomicAdd(&hist[yb], 1u);
        atomicAdd(&hist[zb], 1u);
      }
      row[12] = s >> 32;
      row[13] = 1;
      u64* c = claims + i * 4;
      c[0] = 1;
      c[1] = a;
      c[2] = b;
      c[3] = z;
    } else {
#pragma unroll
      for (int q = 0; q < 14; q++) row[q] = 0;  // padding rows (benches/multi_stark.rs:196-199)
    }
  }
  __syncthreads();
  if (hist[threadIdx.x]) atomicAdd(&byte[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

__global__ void claim_offsets_k(u64* __restrict__ offs, size_t n) {
  const size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (i <= n) offs[i] = 4 * i;
}

}  // namespace

std::unique_ptr<HWitness> witness_u32_add_bench(HSystem& sys, size_t num_adds, u32 a0, u32 b0) {
  Ctx& ctx = *sys.ctx;
  HIP_CHECK(hipSetDevice(ctx.device));
  if (sys.circuits.size() != 2 || sys.circuits[0].main_width != 1 || sys.circuits[0].pre_height != 256 || sys.circuits[1].main_width != 14 ||
      sys.circuits[1].pre_width != 0)
    throw std::runtime_error("witness_u32_add_bench: the system is not [ByteTable, U32Add]");
  if (num_adds == 0 || num_adds > (size_t(1) << NTT_MAX_LOG)) throw std::runtime_error("witness_u32_add_bench: bad size");
  size_t height = 1;
  while (height < num_adds) height <<= 1;
  // powers of the step matrix: column c of T^(2^(j+1)) = T^(2^j) applied to column c of T^(2^j)
  std::unique_ptr<XsJump> jump(new XsJump());
  for (int c = 0; c < 32; c++) jump->m[0][c] = xs_step(1u << c);
  for (int j = 1; j < 32; j++)
    for (int c = 0; c < 32; c++) {
      u32 v = jump->m[j - 1][c], r = 0;
      for (int k = 0; k < 32; k++)
        if ((v >> k) & 1u) r ^= jump->m[j - 1][k];
      jump->m[j][c] = r;
    }
  DBuf<XsJump> d_jump(ctx, 1);
  ctx.h2d(d_jump.p, jump.get(), sizeof(XsJump));
  std::vector<DBuf<u64>> traces(2);
  traces[0] = DBuf<u64>(ctx, 256);
  traces[1] = DBuf<u64>(ctx, height * 14);
  DBuf<u64> d_offs(ctx, num_adds + 1), d_data(ctx, num_adds * 4);
  HIP_CHECK(hipMemsetAsync(traces[0].p, 0, 256 * 8, ctx.stream));
  const size_t threads = (height + GEN_ROWS - 1) / GEN_ROWS;
  hipLaunchKernelGGL(u32_add_bench_k, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx.stream, (const XsJump*)d_jump.p, a0, b0,
                     num_adds, height, traces[1].p, reinterpret_cast<unsigned long long*>(traces[0].p), d_data.p);
  hipLaunchKernelGGL(claim_offsets_k, dim3((unsigned)((num_adds + 256) / 256)), dim3(256), 0, ctx.stream, d_offs.p, num_adds);
  HIP_CHECK(hipGetLastError());
  ctx.sync();  // `jump` may go
  return witness_from_device(sys, std::move(traces), {256, height}, std::move(d_offs), std::move(d_data), num_adds, num_adds * 4);
}

}  // namespace msamd
