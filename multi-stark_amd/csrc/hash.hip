// BLAKE3 Merkle tree build (p3 MerkleTreeMmcs<Goldilocks,u8,SerializingHasher<Blake3>,
// CompressionFunctionFromHasher<Blake3,2,32>,2,32>, /root/reference/src/types.rs:82-83,202-207) and a
// whole-buffer BLAKE3 for the claims part of the Fiat-Shamir transcript (src/prover.rs:369-373).
//
// One thread hashes one row: matrices are column-major, so lane i reads element (i, c) of every column c at
// consecutive addresses (coalesced); each Goldilocks element contributes its canonical value as two
// little-endian 32-bit message words. Shorter matrices are injected at the layer whose length equals their
// height: node = compress(compress(l, r), hash(rows)).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "b3_dev.h"
#include "b3_quad.h"
#include "outer_dev.h"
#include "challenge_dev.h"
#include "fri_dev.h"
#include "msamd.h"
#include "tree_dev.h"

namespace msamd {

namespace {

struct RowIter {
  const MatRef* g;
  size_t H, row;
  unsigned mi;
  u32 c;
  const u64* cur;  // column c of matrix mi at this row
  u32 w;           // width of matrix mi
  size_t stride;   // its column stride (H unless the rows are a window of a taller matrix)
  __device__ __forceinline__ void init() {
    cur = g[0].d + row;
    w = g[0].w;
    stride = g[0].stride;
  }
  __device__ __forceinline__ u64 next() {
    while (c == w) {  // next matrix of the group (never taken for a single-matrix group)
      mi++;
      c = 0;
      cur = g[mi].d + row;
      w = g[mi].w;
      stride = g[mi].stride;
    }
    // the matrix pointers come out of a descriptor in memory, so the compiler only knows them as generic pointers and
    // would emit flat loads (which also count against the LDS counter); they are device allocations: say so
    typedef const u64 __attribute__((address_space(1))) * GlobalPtr;
    u64 v = *(GlobalPtr)cur;
    cur += stride;
    c++;
    return v;
  }
};

__device__ __forceinline__ void parent_cv(const u32 l[8], const u32 r[8], u32 flags_extra, u32 out[8]) {
  u32 m[16];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    m[i] = l[i];
    m[8 + i] = r[i];
  }
  b3_iv(out);
  b3_compress(out, m, 0, 64, B3_PARENT | flags_extra);
}

// BLAKE3 of the serialised row (total_w elements, 8 bytes each). MULTI: rows longer than one 1024-byte chunk.
template <bool MULTI>
__device__ __forceinline__ void hash_row(const MatRef* g, size_t H, size_t row, u32 total_w, u32 out[8]) {
  RowIter it{g, H, row, 0, 0, nullptr, 0, 0};
  it.init();
  u32 cv[8];
  b3_iv(cv);
  u32 m[16];
  u32 stack[MULTI ? 8 * 10 : 8];  // chaining-value stack (up to 2^10 chunks = 1 MiB rows)
  u32 stack_len = 0;
  u64 chunk = 0;
  u32 nvalid = 0;
  // software pipeline: the loads of block b + 1 are issued before block b is compressed
  u64 nxt[8];
  {
    const u32 nv = total_w < 8 ? total_w : 8;
#pragma unroll
    for (u32 j = 0; j < 8; j++) nxt[j] = j < nv ? it.next() : 0;
  }
  for (u32 base = 0; base < total_w; base += 8) {
    nvalid = total_w - base < 8 ? total_w - base : 8;
#pragma unroll
    for (u32 j = 0; j < 8; j++) {
      m[2 * j] = (u32)nxt[j];
      m[2 * j + 1] = (u32)(nxt[j] >> 32);
    }
    if (base + 8 < total_w) {
      const u32 rem = total_w - base - 8;
      const u32 nv = rem < 8 ? rem : 8;
#pragma unroll
      for (u32 j = 0; j < 8; j++) nxt[j] = j < nv ? it.next() : 0;
      // the block in m is full and more input follows
      u32 bic = (base >> 3) & 15;  // index of this block within its chunk
      u32 flags = bic == 0 ? B3_CHUNK_START : 0;
      if (MULTI && bic == 15) {
        b3_compress(cv, m, chunk, 64, flags | B3_CHUNK_END);
        u64 total_chunks = chunk + 1;
        while ((total_chunks & 1) == 0) {
          stack_len--;
          u32 l[8], t[8];
#pragma unroll
          for (int i = 0; i < 8; i++) l[i] = stack[stack_len * 8 + i];
          parent_cv(l, cv, 0, t);
#pragma unroll
          for (int i = 0; i < 8; i++) cv[i] = t[i];
          total_chunks >>= 1;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) stack[stack_len * 8 + i] = cv[i];
        stack_len++;
        chunk++;
        b3_iv(cv);
      } else {
        b3_compress(cv, m, chunk, 64, flags);
      }
    }
  }
  u32 nblocks = (total_w + 7) >> 3;
  u32 bic = (nblocks - 1) & 15;
  u32 flags = (bic == 0 ? B3_CHUNK_START : 0) | B3_CHUNK_END;
  if (!MULTI || stack_len == 0) flags |= B3_ROOT;
  b3_compress(cv, m, chunk, nvalid * 8, flags);
  if (MULTI) {
    while (stack_len > 0) {
      stack_len--;
      u32 l[8], t[8];
#pragma unroll
      for (int i = 0; i < 8; i++) l[i] = stack[stack_len * 8 + i];
      parent_cv(l, cv, stack_len == 0 ? B3_ROOT : 0, t);
#pragma unroll
      for (int i = 0; i < 8; i++) cv[i] = t[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; i++) out[i] = cv[i];
}

template <bool MULTI>
__global__ __launch_bounds__(256) void leaf_hash_k(const MatRef* __restrict__ g, size_t H, u32 total_w, Digest* out) {
  size_t row = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (row >= H) return;
  u32 cv[8];
  hash_row<MULTI>(g, H, row, total_w, cv);
  store_digest(out + row, cv);
}

// The tallest group of a commitment is almost always ONE matrix of at most 128 columns (one BLAKE3 chunk per row). The
// general kernel above walks descriptor lists with data-dependent loops around every load, which leaves the compiler no
// room: the loads sit under branches and everything outstanding is waited for in front of each compression. Here the
// column pointer is a kernel argument, every load is unconditional (the last block re-reads the last column and masks
// it) and the loads of block b + 1 are issued in front of the compression of block b.
__global__ __launch_bounds__(256) void leaf_hash_single_k(const u64* __restrict__ d, size_t rows, size_t H /* column stride */, u32 w,
                                                          Digest* __restrict__ out) {
  const size_t row = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (row >= rows) return;
  const u64* __restrict__ p = d + row;
  const u32 nblocks = (w + 7) >> 3, last_c = w - 1;
  u32 cv[8];
  b3_iv(cv);
  u64 cur[8], nxt[8];
#pragma unroll
  for (u32 j = 0; j < 8; j++) cur[j] = p[size_t(j < last_c ? j : last_c) * H];
  for (u32 b = 0; b + 1 < nblocks; b++) {  // full blocks with more input behind them
    const u32 c0 = 8 * (b + 1);
#pragma unroll
    for (u32 j = 0; j < 8; j++) nxt[j] = p[size_t(c0 + j < last_c ? c0 + j : last_c) * H];
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch in front of the rounds
    u32 m[16];
#pragma unroll
    for (u32 j = 0; j < 8; j++) {
      m[2 * j] = (u32)cur[j];
      m[2 * j + 1] = (u32)(cur[j] >> 32);
    }
    b3_compress(cv, m, 0, 64, b == 0 ? B3_CHUNK_START : 0);
#pragma unroll
    for (u32 j = 0; j < 8; j++) cur[j] = nxt[j];
  }
  const u32 nvalid = w - 8 * (nblocks - 1);
  u32 m[16];
#pragma unroll
  for (u32 j = 0; j < 8; j++) {
    const u64 v = j < nvalid ? cur[j] : 0;
    m[2 * j] = (u32)v;
    m[2 * j + 1] = (u32)(v >> 32);
  }
  b3_compress(cv, m, 0, nvalid * 8, (nblocks == 1 ? B3_CHUNK_START : 0) | B3_CHUNK_END | B3_ROOT);
  store_digest(out + row, cv);
}

// ---- rows far longer than one BLAKE3 chunk on FEW rows (the 2625-column matrix of the reference's BLAKE3 system injected at a
// 1024-row level: 21 chunks per row): a thread per row hashes its 328 blocks one after the other - 390 us of latency on a
// thousand threads. BLAKE3 is a tree hash: one thread per (row, chunk) computes the chunk's chaining value, one thread per
// row then merges them (the stack discipline of hash_row<true>), and the tree kernels read the finished row digests.
__global__ __launch_bounds__(256) void wide_rows_chunk_k(const MatRef* __restrict__ g, size_t H, size_t rows, u32 total_w, u32 nchunks,
                                                         Digest* __restrict__ cvs) {
  const size_t id = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (id >= rows * nchunks) return;
  const size_t row = id / nchunks;
  const u32 chunk = (u32)(id % nchunks);
  const u32 e0 = chunk * 128, e1 = min(total_w, e0 + 128);
  RowIter it{g, H, row, 0, 0, nullptr, 0, 0};
  it.init();
  {  // seek to element e0 of the concatenated row
    u32 skip = e0;
    while (skip >= it.w) {
      skip -= it.w;
      it.mi++;
      it.cur = g[it.mi].d + row;
      it.w = g[it.mi].w;
      it.stride = g[it.mi].stride;
    }
    it.c = skip;
    it.cur += size_t(skip) * it.stride;
  }
  u32 cv[8];
  b3_iv(cv);
  const u32 nblocks = (e1 - e0 + 7) >> 3;
  for (u32 b = 0; b < nblocks; b++) {
    const u32 nv = min(8u, e1 - e0 - 8 * b);
    u32 m[16];
#pragma unroll
    for (u32 j = 0; j < 8; j++) {
      const u64 v = j < nv ? it.next() : 0;
      m[2 * j] = (u32)v;
      m[2 * j + 1] = (u32)(v >> 32);
    }
    b3_compress(cv, m, chunk, nv * 8, (b == 0 ? B3_CHUNK_START : 0u) | (b + 1 == nblocks ? B3_CHUNK_END : 0u));
  }
  store_digest(cvs + id, cv);
}
__global__ __launch_bounds__(256) void wide_rows_merge_k(const Digest* __restrict__ cvs, size_t rows, u32 nchunks, Digest* __restrict__ out) {
  const size_t row = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (row >= rows) return;
  u32 stack[8 * 12];  // sub-tree roots of the chunks so far (2^12 chunks = 4 MiB rows)
  u32 stack_len = 0, cv[8];
  for (u32 c = 0; c + 1 < nchunks; c++) {
    load_digest(cvs + row * nchunks + c, cv);
    for (u32 total = c + 1; (total & 1) == 0; total >>= 1) {
      u32 l[8], t[8];
      stack_len--;
#pragma unroll
      for (int i = 0; i < 8; i++) l[i] = stack[stack_len * 8 + i];
      parent_cv(l, cv, 0, t);
#pragma unroll
      for (int i = 0; i < 8; i++) cv[i] = t[i];
    }
#pragma unroll
    for (int i = 0; i < 8; i++) stack[stack_len * 8 + i] = cv[i];
    stack_len++;
  }
  load_digest(cvs + row * nchunks + (nchunks - 1), cv);
  while (stack_len > 0) {
    u32 l[8], t[8];
    stack_len--;
#pragma unroll
    for (int i = 0; i < 8; i++) l[i] = stack[stack_len * 8 + i];
    parent_cv(l, cv, stack_len == 0 ? B3_ROOT : 0, t);
#pragma unroll
    for (int i = 0; i < 8; i++) cv[i] = t[i];
  }
  store_digest(out + row, cv);
}

// next[i] = compress(prev[2i], prev[2i+1]); with an injected group: compress(that, hash(rows i))
template <bool INJECT, bool MULTI>
__global__ __launch_bounds__(256) void compress_layer_k(const Digest* __restrict__ prev, Digest* __restrict__ next, size_t n,
                                                        const MatRef* __restrict__ g, u32 total_w) {
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  u32 l[8], r[8], d[8];
  load_digest(prev + 2 * i, l);
  load_digest(prev + 2 * i + 1, r);
  b3_compress_pair_root(l, r, d);
  if (INJECT) {
    u32 rh[8], e[8];
    hash_row<MULTI>(g, n, i, total_w, rh);
    b3_compress_pair_root(d, rh, e);
    store_digest(next + i, e);
  } else {
    store_digest(next + i, d);
  }
}


// ---- fused upper levels. Layers are stored back to back (leaf layer first), so the parents of a layer of
// `len` digests start right after it.
// three levels per launch: thread i turns children [8i, 8i+8) into 4 + 2 + 1 ancestors in registers (no idle
// lanes at any level). Its 256 bytes of children are fetched by the whole wave with coalesced 16-byte loads and
// handed over through LDS (lane stride 272 bytes against bank conflicts): a lane reading its own run directly would
// touch 64 cache lines per load instruction and depend on L1 keeping them for its next 15 loads.
__global__ __launch_bounds__(256) void compress3_k(const Digest* __restrict__ child, Digest* __restrict__ l1, Digest* __restrict__ l2,
                                                   Digest* __restrict__ l3, size_t n3) {
  __shared__ __attribute__((aligned(16))) unsigned char stage[4][64 * 272];
  const size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;  // n3 is a multiple of 256 (host-checked)
  const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  {
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(child + (i - lane) * 8);  // the wave's 512 digests
    uint4 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = src[k * 64 + lane];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const u32 piece = k * 64 + lane;  // 16-byte piece index within the wave's block
      *reinterpret_cast<uint4*>(&stage[wave][(piece >> 4) * 272 + (piece & 15) * 16]) = v[k];
    }
  }
  __syncthreads();
  const uint4* mine = reinterpret_cast<const uint4*>(&stage[wave][lane * 272]);
  u32 a[4][8];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint4 p0 = mine[4 * k], p1 = mine[4 * k + 1], p2 = mine[4 * k + 2], p3 = mine[4 * k + 3];
    const u32 l[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
    const u32 r[8] = {p2.x, p2.y, p2.z, p2.w, p3.x, p3.y, p3.z, p3.w};
    b3_compress_pair_root(l, r, a[k]);
    store_digest(l1 + 4 * i + k, a[k]);
  }
  u32 b0[8], b1[8], c[8];
  b3_compress_pair_root(a[0], a[1], b0);
  b3_compress_pair_root(a[2], a[3], b1);
  store_digest(l2 + 2 * i, b0);
  store_digest(l2 + 2 * i + 1, b1);
  b3_compress_pair_root(b0, b1, c);
  store_digest(l3 + i, c);
}

// latency-oriented variant for small layers: 1024 threads turn 2048 children into 1024 + 512 + 256 ancestors with
// one compression per thread per level (three dependent compressions instead of seven)
__global__ __launch_bounds__(1024) void compress3_lds_k(const Digest* __restrict__ child, Digest* __restrict__ l1,
                                                        Digest* __restrict__ l2, Digest* __restrict__ l3) {
  __shared__ __attribute__((aligned(16))) u32 sh1[1024 * 8];
  __shared__ __attribute__((aligned(16))) u32 sh2[512 * 8];
  const u32 t = threadIdx.x;
  const size_t b = blockIdx.x;
  {
    u32 l[8], r[8], d[8];
    load_digest(child + b * 2048 + 2 * t, l);
    load_digest(child + b * 2048 + 2 * t + 1, r);
    b3_compress_pair_root(l, r, d);
    store_digest(l1 + b * 1024 + t, d);
    lds_store_digest(sh1, t, d);
  }
  __syncthreads();
  if (t < 512) {
    u32 l[8], r[8], d[8];
    lds_load_digest(sh1, 2 * t, l);
    lds_load_digest(sh1, 2 * t + 1, r);
    b3_compress_pair_root(l, r, d);
    store_digest(l2 + b * 512 + t, d);
    lds_store_digest(sh2, t, d);
  }
  __syncthreads();
  if (t < 256) {
    u32 l[8], r[8], d[8];
    lds_load_digest(sh2, 2 * t, l);
    lds_load_digest(sh2, 2 * t + 1, r);
    b3_compress_pair_root(l, r, d);
    store_digest(l3 + b * 256 + t, d);
  }
}

// every remaining level of a tree whose current layer has len <= 1024 digests (no injection), one workgroup.
// The levels are a chain of dependent compressions with ever fewer nodes, so each node is computed by a quad
// (b3_quad.h): 256 nodes per pass of the 1024 threads. With CH the launch continues with that FRI round's
// challenger step: the root is observed, the proof-of-work witness searched and beta sampled on the device.
template <bool CH>
__global__ __launch_bounds__(1024) void tree_tail_k(Digest* __restrict__ layer, u32 len, FriChallenge fc) {
  __shared__ __attribute__((aligned(16))) u32 sh[1024 * 8];
  __shared__ ChallengeShared cs;
  const u32 t = threadIdx.x;
  if (t < len) {
    u32 d[8];
    load_digest(layer + t, d);
    lds_store_digest(sh, t, d);
  }
  if (CH && t < 8) cs.st[t] = fc.state[t];
  __syncthreads();
  u32* out = reinterpret_cast<u32*>(layer + len);
  const u32 quad = t >> 2, c = t & 3;
  for (u32 n = len >> 1; n >= 1; n >>= 1) {
    u32 lo[2], hi[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
      const u32 q = quad + 256 * p;
      if (q < n) b3_quad_parent(sh + 16 * q, lo[p], hi[p]);
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; p++) {
      const u32 q = quad + 256 * p;
      if (q < n) {
        sh[8 * q + c] = lo[p];
        sh[8 * q + 4 + c] = hi[p];
        out[8 * q + c] = lo[p];
        out[8 * q + 4 + c] = hi[p];
      }
    }
    __syncthreads();
    out += 8 * n;
  }
  if (CH) {
    challenger_round<1024>(cs, sh, fc.pow_bits);
    if (t < 8) {
      fc.state[t] = cs.st[t];
      fc.rec->root[t] = sh[t];
    }
    if (t == 0) {
      fc.rec->witness = cs.wit;
      fc.rec->beta = cs.beta;
    }
  }
}

// Every level above a layer of `len` digests in ONE launch (len = 2^k, 2 <= len <= 2^21). A workgroup owns `sub`
// consecutive children, i.e. a whole sub-tree. Dependent compressions are latency-bound, so each level picks its form
// by its size (tools/micro/quad_chain.hip: 1.17 us for a one-lane compression, 0.6 us on a quad, but a quad pass over
// all 1024 threads costs 3.2 us): one lane per node while a level has 128 nodes or more, quads (b3_quad.h) below.
// With several workgroups the sub-tree roots are handed over in-launch: each root is written through (agent-scope
// relaxed atomic stores = `sc1`), the workgroup drains its stores and draws a ticket (agent-scope atomic add); the
// workgroup whose ticket is the last one reads all roots with `sc1` loads and computes the remaining levels
// (MI355X_MICROARCH.md, inter-workgroup visibility: write-through payload + drained counter, consumer = last arriver).
// Every layer is still written to global memory for the query phase, none is read back.
// INJ: one injected group (shorter matrices) may sit at any level: that level runs one thread per node.
// CH: the last workgroup continues with the FRI round's challenger step (challenge_dev.h).
// FOLD: the children are the leaf digests of a FRI layer that this launch produces itself: it folds the previous
// layer with the challenge the previous round left on the device, writes the folded layer and hashes its rows - a whole
// commit-phase round (fold, leaves, tree, observe / grind / sample) per launch.
struct SubtreeParams {
  Digest* child;
  u32 len;
  u32 sub;       // children per workgroup: 2^j, 2 <= sub <= 2048; len / sub workgroups (<= 1024)
  u32 inj_len;   // length of the layer that takes the injected group (0 = none)
  u32 inj_w;
  const MatRef* g;
  u32 inj_multi; // rows longer than one BLAKE3 chunk
  u32 pad;
  u32* counter;  // zero at launch; reset by the last workgroup
  FriChallenge fc;
  FoldArgs fold;            // FOLD: out has 2 * len elements, cur 4 * len
  const FriTailRound* prev; // FOLD: the round whose beta folds
  const Digest* inj_hashed; // the injected group's row digests, already computed (wide_rows_*_k); null = hash the rows here
};

// one level of a sub-tree held in LDS: n nodes from 2n children at sh (digest i at sh + 8 i), results back to sh[0 .. n)
// and to gout[0 .. n). Called by all 1024 threads.
template <bool INJ>
__device__ __forceinline__ void tree_level(u32* sh, u32 n, Digest* gout, const SubtreeParams& p, u32 glen, size_t gfirst) {
  const u32 t = threadIdx.x;
  if (INJ && p.inj_len == glen) {  // node = compress(compress(l, r), hash(rows)): one thread per node
    u32 d[8];
    const bool act = t < n;
    if (act) {
      u32 l[8], r[8], e[8], rh[8];
      lds_load_digest(sh, 2 * t, l);
      lds_load_digest(sh, 2 * t + 1, r);
      b3_compress_pair_root(l, r, e);
      if (p.inj_hashed)
        load_digest(p.inj_hashed + (gfirst + t), rh);
      else if (p.inj_multi)
        hash_row<true>(p.g, glen, gfirst + t, p.inj_w, rh);
      else
        hash_row<false>(p.g, glen, gfirst + t, p.inj_w, rh);
      b3_compress_pair_root(e, rh, d);
    }
    __syncthreads();
    if (act) {
      lds_store_digest(sh, t, d);
      store_digest(gout + t, d);
    }
    __syncthreads();
  } else {
    tree_level_plain(sh, n, gout);
  }
}

template <bool CH, bool INJ, bool FOLD>
__global__ __launch_bounds__(1024) void subtree_k(SubtreeParams p) {
  __shared__ __attribute__((aligned(16))) u32 sh[1024 * 8];
  __shared__ ChallengeShared cs;
  __shared__ u32 s_last;
  const u32 t = threadIdx.x, nb = gridDim.x;
  u32 b = blockIdx.x;
  Digest* lvl = p.child + p.len;  // the first parent layer
  u32 glen = p.len >> 1;          // its length
  u32 n;
  E2 hb = e2(0), rf = e2(0);
  if (FOLD) {
    const E2 beta = p.prev->beta;
    hb = e2_mul_base(beta, 0x7FFFFFFF80000001ULL);
    rf = e2_sqr(beta);
  }
  if (p.sub == 2048) {  // level 1 in registers: one node per thread from its two children
    u32 l[8], r[8], d[8];
    const size_t c0 = size_t(b) * 2048 + 2 * t;
    if (FOLD) {
      u32 m[16];
      E2 o0 = fri_fold_one(p.fold, 2 * c0, hb, rf), o1 = fri_fold_one(p.fold, 2 * c0 + 1, hb, rf);
      p.fold.out[2 * c0] = o0;
      p.fold.out[2 * c0 + 1] = o1;
      fri_row_block(o0, o1, m);
      b3_iv(l);
      b3_compress(l, m, 0, 32, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
      o0 = fri_fold_one(p.fold, 2 * c0 + 2, hb, rf);
      o1 = fri_fold_one(p.fold, 2 * c0 + 3, hb, rf);
      p.fold.out[2 * c0 + 2] = o0;
      p.fold.out[2 * c0 + 3] = o1;
      fri_row_block(o0, o1, m);
      b3_iv(r);
      b3_compress(r, m, 0, 32, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
      store_digest(p.child + c0, l);
      store_digest(p.child + c0 + 1, r);
    } else {
      load_digest(p.child + c0, l);
      load_digest(p.child + c0 + 1, r);
    }
    b3_compress_pair_root(l, r, d);
    if (INJ && p.inj_len == glen) {
      u32 rh[8], e[8];
      if (p.inj_hashed)
        load_digest(p.inj_hashed + (size_t(b) * 1024 + t), rh);
      else if (p.inj_multi)
        hash_row<true>(p.g, glen, size_t(b) * 1024 + t, p.inj_w, rh);
      else
        hash_row<false>(p.g, glen, size_t(b) * 1024 + t, p.inj_w, rh);
      b3_compress_pair_root(d, rh, e);
#pragma unroll
      for (int k = 0; k < 8; k++) d[k] = e[k];
    }
    store_digest(lvl + size_t(b) * 1024 + t, d);
    lds_store_digest(sh, t, d);
    lvl += glen;
    glen >>= 1;
    n = 512;
  } else {  // sub <= 1024 children straight into LDS
    if (t < p.sub) {
      u32 d[8];
      const size_t c0 = size_t(b) * p.sub + t;
      if (FOLD) {
        u32 m[16];
        const E2 o0 = fri_fold_one(p.fold, 2 * c0, hb, rf), o1 = fri_fold_one(p.fold, 2 * c0 + 1, hb, rf);
        p.fold.out[2 * c0] = o0;
        p.fold.out[2 * c0 + 1] = o1;
        fri_row_block(o0, o1, m);
        b3_iv(d);
        b3_compress(d, m, 0, 32, B3_CHUNK_START | B3_CHUNK_END | B3_ROOT);
        store_digest(p.child + c0, d);
      } else {
        load_digest(p.child + c0, d);
      }
      lds_store_digest(sh, t, d);
    }
    n = p.sub >> 1;
  }
  if (CH && t < 8) cs.st[t] = p.fc.state[t];
  __syncthreads();
#pragma unroll 1
  for (int phase = 0; phase < 2; phase++) {
    if (phase == 1) {
      if (nb == 1) break;
      // hand the sub-tree root (sh[0..7], already stored plainly for later kernels) to the last workgroup
      Digest* roots = lvl - nb;  // the layer of nb digests this phase just completed
      if (t < 8) __hip_atomic_store(reinterpret_cast<u32*>(roots + b) + t, sh[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) {
        const u32 ticket = __hip_atomic_fetch_add(p.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = ticket == nb - 1 ? 1u : 0u;
      }
      __syncthreads();
      if (!s_last) return;
      const u32* rw = reinterpret_cast<const u32*>(roots);
      for (u32 i = t; i < nb * 8; i += 1024) sh[i] = __hip_atomic_load(rw + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (t == 0) __hip_atomic_store(p.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      b = 0;
      n = nb >> 1;
    }
#pragma unroll 1
    for (; n >= 1; n >>= 1) {
      tree_level<INJ>(sh, n, lvl + size_t(b) * n, p, glen, size_t(b) * n);
      lvl += glen;
      glen >>= 1;
    }
  }
  if (CH) {
    challenger_round<1024>(cs, sh, p.fc.pow_bits);
    if (t < 8) {
      p.fc.state[t] = cs.st[t];
      p.fc.rec->root[t] = sh[t];
    }
    if (t == 0) {
      p.fc.rec->witness = cs.wit;
      p.fc.rec->beta = cs.beta;
    }
  }
}

// ---- whole-stream BLAKE3. The stream is `prefix` (prefix_len bytes, any alignment) followed by `nwords`
// little-endian u64 words (8-byte aligned), which is how the transcript up to the claims is shaped: a short
// host-built prefix, then the length-prefixed claims as field elements.
__device__ __forceinline__ u32 stream_byte(const uint8_t* __restrict__ prefix, size_t pl, const u64* __restrict__ words, size_t p) {
  if (p < pl) return prefix[p];
  size_t q = p - pl;
  return (u32)((words[q >> 3] >> (8 * (q & 7))) & 0xff);
}

__global__ __launch_bounds__(256) void chunk_cv_k(const uint8_t* __restrict__ prefix, size_t pl, const u64* __restrict__ words,
                                                  size_t len, size_t nchunks, size_t c0, size_t c1, Digest* out) {
  size_t ch = c0 + blockIdx.x * size_t(blockDim.x) + threadIdx.x;  // chunks [c0, c1) of the stream's nchunks
  if (ch >= c1) return;
  size_t off = ch * 1024;
  size_t clen = len - off < 1024 ? len - off : 1024;
  u32 nblocks = clen == 0 ? 1 : (u32)((clen + 63) / 64);
  u32 cv[8];
  b3_iv(cv);
  const bool single = nchunks == 1;
  for (u32 b = 0; b < nblocks; b++) {
    u32 m[16];
    size_t boff = off + size_t(b) * 64;
    u32 bl = (u32)(clen - size_t(b) * 64 < 64 ? clen - size_t(b) * 64 : 64);
    if (bl == 64 && boff >= pl) {
      size_t q = boff - pl;
      const u64* w = words + (q >> 3);
      unsigned s = (unsigned)(q & 7) * 8;
      u64 cur = w[0];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        u64 x;
        if (s == 0) {
          x = cur;
          if (k < 7) cur = w[k + 1];
        } else {
          u64 nxt = w[k + 1];
          x = (cur >> s) | (nxt << (64 - s));
          cur = nxt;
        }
        m[2 * k] = (u32)x;
        m[2 * k + 1] = (u32)(x >> 32);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        u32 wv = 0;
        for (int k = 0; k < 4; k++) {
          u32 idx = 4 * i + k;
          u32 byte = idx < bl ? stream_byte(prefix, pl, words, boff + idx) : 0;
          wv |= byte << (8 * k);
        }
        m[i] = wv;
      }
    }
    u32 flags = (b == 0 ? B3_CHUNK_START : 0) | (b == nblocks - 1 ? (B3_CHUNK_END | (single ? B3_ROOT : 0)) : 0);
    b3_compress(cv, m, ch, bl, flags);
  }
  store_digest(out + ch, cv);
}

// The chunks that hold prefix bytes (normally chunk 0 alone): their bytes come from two sources at byte granularity, which
// in chunk_cv_k is a byte-wise load path - one lane walking it made that lane the longest of the whole launch (58 us).
// Here a wave assembles the chunk in LDS (four bytes per lane and step) and lane 0 runs the 16 dependent compressions.
__global__ __launch_bounds__(64) void chunk_cv_prefix_k(const uint8_t* __restrict__ prefix, size_t pl, const u64* __restrict__ words,
                                                        size_t len, size_t nchunks, size_t c0, Digest* out) {
  __shared__ u32 sh[256];
  const size_t ch = c0 + blockIdx.x;
  const size_t off = ch * 1024;
  const size_t clen = len - off < 1024 ? len - off : 1024;
  for (u32 i = threadIdx.x; i < 256; i += 64) {
    u32 wv = 0;
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
      const u32 idx = 4 * i + k;
      const u32 byte = idx < clen ? stream_byte(prefix, pl, words, off + idx) : 0;
      wv |= byte << (8 * k);
    }
    sh[i] = wv;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const u32 nblocks = clen == 0 ? 1 : (u32)((clen + 63) / 64);
  u32 cv[8];
  b3_iv(cv);
  const bool single = nchunks == 1;
  for (u32 b = 0; b < nblocks; b++) {
    u32 m[16];
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = sh[16 * b + i];
    const u32 bl = (u32)(clen - size_t(b) * 64 < 64 ? clen - size_t(b) * 64 : 64);
    const u32 flags = (b == 0 ? B3_CHUNK_START : 0) | (b == nblocks - 1 ? (B3_CHUNK_END | (single ? B3_ROOT : 0)) : 0);
    b3_compress(cv, m, ch, bl, flags);
  }
  store_digest(out + ch, cv);
}

// one level of the left-full tree: pair adjacent chaining values, carry an odd last one up unchanged
__global__ __launch_bounds__(256) void cv_level_k(const Digest* __restrict__ prev, Digest* __restrict__ next, size_t n_prev,
                                                  u32 root_flag) {
  size_t n_next = (n_prev + 1) / 2;
  size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (i >= n_next) return;
  u32 l[8];
  load_digest(prev + 2 * i, l);
  if (2 * i + 1 < n_prev) {
    u32 r[8], o[8];
    load_digest(prev + 2 * i + 1, r);
    parent_cv(l, r, root_flag, o);
    store_digest(next + i, o);
  } else {
    store_digest(next + i, l);
  }
}

// the last levels of the left-full tree (n <= 2048 chaining values) in one workgroup, one parent per quad: replaces a
// dozen launches of cv_level_k whose grids have shrunk to a few hundred threads
// (sib_out: also write, per level, the value at index 1 BEFORE the level is paired - the sibling of the leftmost path)
__global__ __launch_bounds__(1024) void cv_tail_k(const Digest* __restrict__ prev, Digest* __restrict__ out, u32 n, Digest* __restrict__ sib_out = nullptr) {
  __shared__ __attribute__((aligned(16))) u32 sh[2048 * 8];
  const u32 t = threadIdx.x;
  for (u32 i = t; i < n; i += 1024) {
    u32 d[8];
    load_digest(prev + i, d);
    lds_store_digest(sh, i, d);
  }
  __syncthreads();
  const u32 quad = t >> 2, c = t & 3;
  while (n > 1) {
    const u32 nn = (n + 1) / 2;
    const u32 flags = B3_PARENT | (n == 2 ? (u32)B3_ROOT : 0u);
    if (sib_out) {
      if (t < 8) reinterpret_cast<u32*>(sib_out)[t] = sh[8 + t];
      sib_out++;
    }
    u32 lo[4], hi[4];
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
      const u32 q = quad + 256 * ps;
      if (q < nn) {
        if (2 * q + 1 < n) {
          b3_quad_compress_iv<false>(sh + 16 * q, 64, flags, lo[ps], hi[ps]);
        } else {  // an odd last value moves up unchanged
          lo[ps] = sh[16 * q + c];
          hi[ps] = sh[16 * q + 4 + c];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
      const u32 q = quad + 256 * ps;
      if (q < nn) {
        sh[8 * q + c] = lo[ps];
        sh[8 * q + 4 + c] = hi[ps];
      }
    }
    __syncthreads();
    n = nn;
  }
  if (t < 8) reinterpret_cast<u32*>(out)[t] = sh[t];
}

}  // namespace

void merkle_alloc(Ctx& ctx, DTree& t, size_t maxh) {
  t.layer_off.clear();
  t.layer_len.clear();
  size_t tot = 0;
  for (size_t l = maxh;; l >>= 1) {
    t.layer_off.push_back(tot);
    t.layer_len.push_back(l);
    tot += l;
    if (l == 1) break;
  }
  t.digests = DBuf<Digest>(ctx, tot);
}

// levels li_begin.. of tree t, given which layers receive an injected group (inject[li] = (first, total_w, count))
struct InjectAt {
  size_t first = 0;
  u32 total_w = 0;
  size_t count = 0;
};
// children per workgroup of subtree_k: one workgroup up to 1024 children; above that about 256 workgroups (one per CU),
// each owning between 8 and 2048 children, so that the leaf work spreads over the chip while the hand-over stays one hop
static u32 subtree_children_per_group(size_t len) {
  if (len <= 1024) return (u32)len;
  size_t sub = len / 256;
  if (sub < 8) sub = 8;
  if (sub > 2048) sub = 2048;
  return (u32)sub;
}

static void build_levels(Ctx& ctx, DTree& t, const std::vector<InjectAt>& inj, const MatRef* drefs, const FriChallenge* fc = nullptr) {
  const size_t L = t.layer_len.size();
  size_t last_inject = 0;
  for (size_t li = 1; li < L; li++)
    if (inj[li].count) last_inject = li;
  size_t li = 1;
  const bool no_subtree = getenv("MSAMD_NO_SUBTREE") != nullptr;  // read at call time (tests flip it)
  const char* sml = getenv("MSAMD_SUBTREE_MAX_LOG");
  const unsigned subtree_max_log = sml ? (unsigned)atoi(sml) : 20u;
  while (li < L) {
    const size_t child_len = t.layer_len[li - 1];
    Digest* child = t.base() + t.layer_off[li - 1];
    // all remaining levels in one launch (sub-trees per workgroup + in-launch hand-over of their roots)
    if (!no_subtree && child_len >= 2 && child_len <= (size_t(1) << std::min(subtree_max_log, 21u))) {
      size_t n_inj = 0, inj_li = 0;
      for (size_t k = li; k < L; k++)
        if (inj[k].count) {
          n_inj++;
          inj_li = k;
        }
      if (n_inj == 0 || (n_inj == 1 && !fc)) {
        SubtreeParams sp;
        memset(&sp, 0, sizeof(sp));
        sp.child = child;
        sp.len = (u32)child_len;
        sp.sub = subtree_children_per_group(child_len);
        sp.inj_len = n_inj ? (u32)t.layer_len[inj_li] : 0u;
        sp.g = n_inj ? drefs + inj[inj_li].first : nullptr;
        sp.inj_w = n_inj ? inj[inj_li].total_w : 0u;
        sp.inj_multi = n_inj && inj[inj_li].total_w > 128 ? 1u : 0u;
        // few rows of many chunks each: the row digests are computed chunk-parallel in front of the tree launch
        DBuf<Digest> wide_cvs, wide_rows;
        if (sp.inj_multi && sp.inj_len <= 16384 && sp.inj_w >= 512 && !getenv("MSAMD_NO_WIDE_PREHASH")) {
          const size_t rows = sp.inj_len;
          const u32 nchunks = (sp.inj_w + 127) / 128;
          wide_cvs = DBuf<Digest>(ctx, rows * nchunks);
          wide_rows = DBuf<Digest>(ctx, rows);
          hipEvent_t evw = ctx.prof_begin(K_LEAF_HASH);
          hipLaunchKernelGGL(wide_rows_chunk_k, dim3((unsigned)((rows * nchunks + 255) / 256)), dim3(256), 0, ctx.stream, sp.g, rows, rows, sp.inj_w,
                             nchunks, wide_cvs.p);
          hipLaunchKernelGGL(wide_rows_merge_k, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx.stream, wide_cvs.p, rows, nchunks, wide_rows.p);
          ctx.prof_end(K_LEAF_HASH, evw, 8.0 * sp.inj_w * rows);
          HIP_CHECK(hipGetLastError());
          sp.inj_hashed = wide_rows.p;
        }
        sp.counter = tree_counter_slot(ctx);
        sp.fc = fc ? *fc : FriChallenge{};
        const unsigned nb = (unsigned)(child_len / sp.sub);
        const KernelId kid = fc ? K_OTHER : K_COMPRESS;
        hipEvent_t ev = ctx.prof_begin(kid);
        if (fc)
          hipLaunchKernelGGL((subtree_k<true, false, false>), dim3(nb), dim3(1024), 0, ctx.stream, sp);
        else if (n_inj)
          hipLaunchKernelGGL((subtree_k<false, true, false>), dim3(nb), dim3(1024), 0, ctx.stream, sp);
        else
          hipLaunchKernelGGL((subtree_k<false, false, false>), dim3(nb), dim3(1024), 0, ctx.stream, sp);
        ctx.prof_end(kid, ev, 96.0 * double(child_len) + (n_inj ? 8.0 * sp.inj_w * sp.inj_len : 0.0));
        fc = nullptr;
        li = L;
        break;
      }
    }
    if (child_len <= 1024 && li > last_inject) {
      const KernelId kid = fc ? K_OTHER : K_COMPRESS;  // the challenger step's grinding is not tree work
      hipEvent_t ev = ctx.prof_begin(kid);
      if (fc)
        hipLaunchKernelGGL(tree_tail_k<true>, dim3(1), dim3(1024), 0, ctx.stream, child, (u32)child_len, *fc);
      else
        hipLaunchKernelGGL(tree_tail_k<false>, dim3(1), dim3(1024), 0, ctx.stream, child, (u32)child_len, FriChallenge{});
      ctx.prof_end(kid, ev, 96.0 * double(child_len));
      fc = nullptr;
      break;
    }
    if (child_len >= 2048 && li + 2 < L && !inj[li].count && !inj[li + 1].count && !inj[li + 2].count) {
      hipEvent_t ev = ctx.prof_begin(K_COMPRESS);
      const size_t n3 = child_len / 8;
      if (n3 >= (size_t(1) << 17) && n3 % 256 == 0)  // throughput-bound; below that the dependent chain is what costs
        hipLaunchKernelGGL(compress3_k, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, ctx.stream, (const Digest*)child,
                           t.base() + t.layer_off[li], t.base() + t.layer_off[li + 1], t.base() + t.layer_off[li + 2], n3);
      else
        hipLaunchKernelGGL(compress3_lds_k, dim3((unsigned)(child_len / 2048)), dim3(1024), 0, ctx.stream, (const Digest*)child,
                           t.base() + t.layer_off[li], t.base() + t.layer_off[li + 1], t.base() + t.layer_off[li + 2]);
      ctx.prof_end(K_COMPRESS, ev, 32.0 * double(child_len) * 1.875);
      li += 3;
      continue;
    }
    const size_t n = t.layer_len[li];
    Digest* next = t.base() + t.layer_off[li];
    dim3 grid((unsigned)((n + 255) / 256));
    hipEvent_t ev = ctx.prof_begin(K_COMPRESS);
    if (inj[li].count == 0)
      hipLaunchKernelGGL((compress_layer_k<false, false>), grid, dim3(256), 0, ctx.stream, (const Digest*)child, next, n,
                         (const MatRef*)nullptr, 0u);
    else if (inj[li].total_w <= 128)
      hipLaunchKernelGGL((compress_layer_k<true, false>), grid, dim3(256), 0, ctx.stream, (const Digest*)child, next, n,
                         drefs + inj[li].first, inj[li].total_w);
    else
      hipLaunchKernelGGL((compress_layer_k<true, true>), grid, dim3(256), 0, ctx.stream, (const Digest*)child, next, n,
                         drefs + inj[li].first, inj[li].total_w);
    ctx.prof_end(K_COMPRESS, ev, double(n) * (96.0 + 8.0 * inj[li].total_w));
    li++;
  }
  if (fc && L == 1 && li == 1) {  // a single leaf is its own root
    hipLaunchKernelGGL(tree_tail_k<true>, dim3(1), dim3(1024), 0, ctx.stream, t.base(), 1u, *fc);
    fc = nullptr;
  }
  if (fc) throw std::runtime_error("build_levels: the tree never reached the single-workgroup tail");
  HIP_CHECK(hipGetLastError());
}

void merkle_top_challenge(Ctx& ctx, Digest* layer, size_t len, const FriChallenge& fc) {
  if (len < 2 || len > 1024 || (len & (len - 1))) throw std::runtime_error("merkle_top_challenge: 2 .. 1024 roots, a power of two");
  if (!fc.state || !fc.rec) throw std::runtime_error("merkle_top_challenge: challenger state missing");
  hipLaunchKernelGGL(tree_tail_k<true>, dim3(1), dim3(1024), 0, ctx.stream, layer, (u32)len, fc);
  HIP_CHECK(hipGetLastError());
}

void merkle_compress_plain(Ctx& ctx, DTree& t, const FriChallenge* fc) {
  std::vector<InjectAt> inj(t.layer_len.size());
  build_levels(ctx, t, inj, nullptr, fc);
}

bool fri_round_fusable(size_t rows) {
  const char* off = getenv("MSAMD_NO_FRI_FUSED");
  // above 2^MSAMD_FRI_FUSED_MAX_LOG leaves a round is work, not latency: fold + leaf digests, three tree levels per launch
  // and the sub-tree launch as separate kernels keep the chip full where the one-launch form leaves most lanes idle in the
  // upper levels of each workgroup's sub-tree
  const char* ml = getenv("MSAMD_FRI_FUSED_MAX_LOG");
  const unsigned max_log = ml ? (unsigned)atoi(ml) : 21u;
  return !off && !getenv("MSAMD_NO_SUBTREE") && rows >= 4 && rows / 2 <= (size_t(1) << std::min(max_log, 21u));
}

// One commit-phase round in one launch: fold `cur` (2 * rows elements) with the beta of `prev` into `out` (rows elements),
// hash the rows / 2 leaves of the folded layer into t's leaf layer, build t and run the challenger step `fc` on its root.
void fri_round_fused(Ctx& ctx, DTree& t, const E2* cur, size_t rows, const FriTailRound* prev, const E2* roll_in, E2* out,
                     const FriChallenge& fc) {
  const size_t leaves = rows / 2;
  const unsigned lr = log2_strict(rows);
  if (lr + 1 > TW_LOG) throw std::runtime_error("FRI layer above 2^28 is not supported");
  if (!t.digests.p) merkle_alloc(ctx, t, leaves);
  SubtreeParams sp;
  memset(&sp, 0, sizeof(sp));
  sp.child = t.base();
  sp.len = (u32)leaves;
  sp.sub = subtree_children_per_group(leaves);
  sp.counter = tree_counter_slot(ctx);
  sp.fc = fc;
  sp.fold.cur = cur;
  sp.fold.roll = roll_in;
  sp.fold.out = out;
  sp.fold.t0i = ctx.tw0i;
  sp.fold.t1i = ctx.tw1i;
  sp.fold.log_rows = lr;
  sp.prev = prev;
  hipEvent_t ev = ctx.prof_begin(K_FRI_FOLD);
  hipLaunchKernelGGL((subtree_k<true, false, true>), dim3((unsigned)(leaves / sp.sub)), dim3(1024), 0, ctx.stream, sp);
  ctx.prof_end(K_FRI_FOLD, ev, 48.0 * rows + 64.0 * leaves);
  HIP_CHECK(hipGetLastError());
}

void merkle_build(Ctx& ctx, DTree& t) {
  size_t nm = t.mat_d.size();
  if (nm == 0) throw std::runtime_error("merkle_build: no matrices");
  std::vector<size_t> order(nm);
  for (size_t i = 0; i < nm; i++) {
    order[i] = i;
    if (t.mat_h[i] == 0 || (t.mat_h[i] & (t.mat_h[i] - 1))) throw std::runtime_error("merkle_build: heights must be powers of two");
    if (t.mat_w[i] == 0 || t.mat_w[i] > 0xFFFFFFFFu) throw std::runtime_error("merkle_build: bad width");
  }
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return t.mat_h[a] > t.mat_h[b]; });
  size_t maxh = t.mat_h[order[0]];
  merkle_alloc(ctx, t, maxh);
  // group descriptors, sorted order
  std::vector<MatRef> refs(nm);
  if (!t.mat_stride.empty() && t.mat_stride.size() != nm) throw std::runtime_error("merkle_build: one column stride per matrix expected");
  auto stride_of = [&](size_t i) { return t.mat_stride.empty() ? t.mat_h[i] : t.mat_stride[i]; };
  for (size_t k = 0; k < nm; k++) {
    if (stride_of(order[k]) < t.mat_h[order[k]]) throw std::runtime_error("merkle_build: column stride below the matrix height");
    refs[k] = MatRef{t.mat_d[order[k]], (uint32_t)t.mat_w[order[k]], 0, (uint64_t)stride_of(order[k])};
  }
  DBuf<MatRef> drefs(ctx, nm);
  ctx.h2d(drefs.p, refs.data(), nm * sizeof(MatRef));
  size_t pos = 0;
  auto take_group = [&](size_t height, size_t& first, u32& total_w, size_t& count) {
    first = pos;
    total_w = 0;
    count = 0;
    while (pos < nm && t.mat_h[order[pos]] == height) {
      total_w += (u32)t.mat_w[order[pos]];
      pos++;
      count++;
    }
  };
  size_t first, count;
  u32 tw;
  take_group(maxh, first, tw, count);
  {
    // descriptor list must end at the group's end: RowIter walks by widths, total_w bounds it
    dim3 grid((unsigned)((maxh + 255) / 256));
    hipEvent_t ev = ctx.prof_begin(K_LEAF_HASH);
    if (tw <= 128 && count == 1 && !getenv("MSAMD_GENERIC_LEAF_HASH"))
      hipLaunchKernelGGL(leaf_hash_single_k, grid, dim3(256), 0, ctx.stream, t.mat_d[order[first]], maxh, stride_of(order[first]), tw, t.base());
    else if (tw <= 128)
      hipLaunchKernelGGL(leaf_hash_k<false>, grid, dim3(256), 0, ctx.stream, drefs.p + first, maxh, tw, t.base());
    else
      hipLaunchKernelGGL(leaf_hash_k<true>, grid, dim3(256), 0, ctx.stream, drefs.p + first, maxh, tw, t.base());
    ctx.prof_end(K_LEAF_HASH, ev, double(maxh) * (8.0 * tw + 32.0));
  }
  std::vector<InjectAt> inj(t.layer_len.size());
  for (size_t li = 1; li < t.layer_len.size(); li++) {
    take_group(t.layer_len[li], first, tw, count);
    inj[li].first = first;
    inj[li].total_w = tw;
    inj[li].count = count;
  }
  build_levels(ctx, t, inj, drefs.p);
  if (pos != nm) throw std::runtime_error("merkle_build: matrix height not reached");
  HIP_CHECK(hipGetLastError());
}

// The levels ABOVE n sub-tree roots (n a power of two, one root per rank of a joint proof: prover_sharded.inc) hashed where the
// roots are - on the device - so that the commitment never has to visit the host before the transcript step that consumes it:
// out[0 .. n) = the roots, then n / 2 parents, ..., the root (2 n - 1 digests). One workgroup; log2 n dependent compressions.
namespace {
__global__ __launch_bounds__(256) void tree_top_k(const Digest* __restrict__ roots, u32 n, Digest* __restrict__ out) {
  const u32 t = threadIdx.x;
  for (u32 i = t; i < n; i += blockDim.x) {
    u32 d[8];
    load_digest(roots + i, d);
    store_digest(out + i, d);
  }
  __syncthreads();
  u32 off = 0;
  for (u32 len = n; len > 1; len >>= 1) {
    for (u32 i = t; i < len / 2; i += blockDim.x) {
      u32 l[8], r[8], d[8];
      load_digest(out + off + 2 * i, l);
      load_digest(out + off + 2 * i + 1, r);
      b3_compress_pair_root(l, r, d);
      store_digest(out + off + len + i, d);
    }
    __syncthreads();  // (one workgroup: the barrier also orders its global stores and loads)
    off += len;
  }
}
}  // namespace
void merkle_tree_top(Ctx& ctx, const Digest* d_roots, size_t n, Digest* d_out) {
  if (n == 0 || (n & (n - 1)) || n > (size_t(1) << 20)) throw std::runtime_error("merkle_tree_top: root count must be a power of two");
  hipLaunchKernelGGL(tree_top_k, dim3(1), dim3(256), 0, ctx.stream, d_roots, (u32)n, d_out);
  HIP_CHECK(hipGetLastError());
}

std::vector<Digest> merkle_cap(Ctx& ctx, const DTree& t) {
  size_t cl = t.cap_layer();
  std::vector<Digest> cap(t.layer_len[cl]);
  ctx.d2h(cap.data(), t.base() + t.layer_off[cl], cap.size() * sizeof(Digest));
  return cap;
}

size_t blake3_num_chunks(size_t prefix_len, size_t nwords) {
  size_t len = prefix_len + 8 * nwords;
  return len == 0 ? 1 : (len + 1023) / 1024;
}

void blake3_chunk_cvs(Ctx& ctx, const uint8_t* d_prefix, size_t prefix_len, const u64* d_words, size_t nwords, size_t c0, size_t c1,
                      Digest* cvs) {
  const size_t len = prefix_len + 8 * nwords, nchunks = blake3_num_chunks(prefix_len, nwords);
  if (c1 > nchunks) c1 = nchunks;
  if (c0 >= c1) return;
  const size_t pc = std::min(c1, (prefix_len + 1023) / 1024);  // chunks below pc hold prefix bytes
  if (c0 < pc) {
    hipLaunchKernelGGL(chunk_cv_prefix_k, dim3((unsigned)(pc - c0)), dim3(64), 0, ctx.stream, d_prefix, prefix_len, d_words, len, nchunks, c0, cvs);
    c0 = pc;
  }
  if (c0 < c1)
    hipLaunchKernelGGL(chunk_cv_k, dim3((unsigned)((c1 - c0 + 255) / 256)), dim3(256), 0, ctx.stream, d_prefix, prefix_len, d_words, len,
                       nchunks, c0, c1, cvs);
  HIP_CHECK(hipGetLastError());
}

// ---- the tree over the chunks of a stream whose chunk 0 arrives LATE (the claims transcript: chunk 0 holds the stage-1
// commitment). The chaining value of chunk 0 enters the tree on its leftmost path only; the sibling of that path at level l
// is the sub-tree over chunks [2^l, 2^(l+1)) (cut at the end of the stream), none of which depends on chunk 0. They are
// computed early - the ordinary level launches run over all chaining values with a placeholder in slot 0 and every level's
// value at index 1 is kept (beside the stage-1 tree) - and what is left behind the commitment
// is one wave: the 16 blocks of chunk 0 and one parent per level, each compression on a quad (claims_root_k) - instead of
// the chunk, five tree levels and the tree's tail as launches of their own.
struct ClaimsRootArgs {
  const uint8_t* prefix;  // the stream's host-built prefix, placeholder bytes where the commitment goes
  const u64* words;       // the stream behind the prefix
  size_t pl, len;         // prefix length, stream length in bytes
  const u32* cap;         // the commitment (device), patched over prefix bytes [cap_off, cap_off + cap_bytes)
  u32 cap_off, cap_bytes;
  const Digest* sib[32];  // per tree level: where the leftmost path's sibling lies
  u32 nchunks;
  Digest* out;
  ChallengeBG* bg;  // when set: the launch goes on to sample beta and gamma from the digest (outer_dev.h)
  u32* state_out;
};
__global__ __launch_bounds__(64) void claims_root_k(ClaimsRootArgs a) {
  __shared__ u32 sh[256];
  __shared__ u32 pm[16];
  const u32 t = threadIdx.x, c = t & 3;
  const u32 clen = a.len < 1024 ? (u32)a.len : 1024u;
  for (u32 i = t; i < 256; i += 64) {
    u32 wv = 0;
#pragma unroll
    for (u32 k = 0; k < 4; k++) {
      const u32 idx = 4 * i + k;
      u32 byte = 0;
      if (idx < clen) {
        if (idx >= a.cap_off && idx - a.cap_off < a.cap_bytes) {
          const u32 q = idx - a.cap_off;
          byte = (a.cap[q >> 2] >> (8 * (q & 3))) & 0xff;
        } else {
          byte = stream_byte(a.prefix, a.pl, a.words, idx);
        }
      }
      wv |= byte << (8 * k);
    }
    sh[i] = wv;
  }
  __syncthreads();
  // every quad of the wave runs the same chain (the DPP moves stay inside a quad); quad 0 writes
  u32 lo = c == 0 ? B3_IV0 : c == 1 ? B3_IV1 : c == 2 ? B3_IV2 : B3_IV3;
  u32 hi = c == 0 ? B3_IV4 : c == 1 ? B3_IV5 : c == 2 ? B3_IV6 : B3_IV7;
  const u32 nblocks = (clen + 63) / 64;
  for (u32 b = 0; b < nblocks; b++) {
    const u32 bl = clen - 64 * b < 64 ? clen - 64 * b : 64;
    const u32 flags = (b == 0 ? (u32)B3_CHUNK_START : 0u) | (b == nblocks - 1 ? (u32)B3_CHUNK_END : 0u);
    b3_quad_compress_cv(sh + 16 * b, bl, flags, lo, hi);
  }
  for (u32 l = 0; (1u << l) < a.nchunks; l++) {
    if (t < 4) {
      pm[c] = lo;
      pm[4 + c] = hi;
    }
    if (t >= 8 && t < 16) pm[t] = reinterpret_cast<const u32*>(a.sib[l])[t - 8];
    __syncthreads();
    const bool last = (2u << l) >= a.nchunks;
    b3_quad_compress_iv<false>(pm, 64, B3_PARENT | (last ? (u32)B3_ROOT : 0u), lo, hi);
    __syncthreads();
  }
  if (t < 4) {
    u32* o = reinterpret_cast<u32*>(a.out);
    o[c] = lo;
    o[4 + c] = hi;
    pm[c] = lo;
    pm[4 + c] = hi;
  }
  if (a.bg) {
    __syncthreads();
    outer_beta_gamma_step(pm, a.bg, a.state_out);
  }
}

// the tree over the chunks' chaining values; returns where the digest lies on the device (inside cvs or scratch: copied to
// out_dev when given)
static const Digest* blake3_cv_tree(Ctx& ctx, Digest* cvs, size_t nchunks, DBuf<Digest>& scratch, Digest* out_dev) {
  scratch = DBuf<Digest>(ctx, (nchunks + 1) / 2);
  Digest* cur = cvs;
  Digest* nxt = scratch.p;
  size_t n = nchunks;
  while (n > 1) {
    if (n <= 2048) {  // the rest in one launch
      hipLaunchKernelGGL(cv_tail_k, dim3(1), dim3(1024), 0, ctx.stream, (const Digest*)cur, nxt, (u32)n);
      std::swap(cur, nxt);
      break;
    }
    size_t nn = (n + 1) / 2;
    hipLaunchKernelGGL(cv_level_k, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, ctx.stream, cur, nxt, n,
                       n == 2 ? (u32)B3_ROOT : 0u);
    std::swap(cur, nxt);
    n = nn;
  }
  HIP_CHECK(hipGetLastError());
  if (out_dev) HIP_CHECK(hipMemcpyAsync(out_dev, cur, sizeof(Digest), hipMemcpyDeviceToDevice, ctx.stream));
  return cur;
}
Digest blake3_from_cvs(Ctx& ctx, Digest* cvs, size_t nchunks) {
  DBuf<Digest> b;
  const Digest* cur = blake3_cv_tree(ctx, cvs, nchunks, b, nullptr);
  Digest out;
  ctx.d2h(&out, cur, sizeof(Digest));
  return out;
}
// launches only: the digest is left at out_dev (device memory)
void blake3_from_cvs_async(Ctx& ctx, Digest* cvs, size_t nchunks, Digest* out_dev) {
  DBuf<Digest> b;
  (void)blake3_cv_tree(ctx, cvs, nchunks, b, out_dev);
}

// early half: cvs[1 .. nchunks) are complete, cvs[0] is a placeholder. Runs the tree's levels and records where the leftmost
// path's sibling of every level lies (levels: every level in a buffer of its own, kept until the late half has run)
void blake3_late_chunk0_prepare(Ctx& ctx, const Digest* cvs, size_t nchunks, LateChunk0& lc) {
  if (nchunks < 2 || nchunks > (size_t(1) << 30)) throw std::runtime_error("blake3_late_chunk0: chunk count out of range");
  lc.levels = DBuf<Digest>(ctx, nchunks + 64);
  lc.sib.clear();
  const Digest* cur = cvs;
  Digest* nxt = lc.levels.p;
  size_t n = nchunks;
  while (n > 1) {
    if (n <= 2048) {  // the rest in one launch, which writes its levels' siblings one after the other
      Digest* tail_sib = nxt + 1;
      size_t m = n;
      while (m > 1) {
        lc.sib.push_back(tail_sib++);
        m = (m + 1) / 2;
      }
      hipLaunchKernelGGL(cv_tail_k, dim3(1), dim3(1024), 0, ctx.stream, cur, nxt, (u32)n, nxt + 1);
      break;
    }
    lc.sib.push_back(cur + 1);
    const size_t nn = (n + 1) / 2;
    hipLaunchKernelGGL(cv_level_k, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, ctx.stream, cur, nxt, n, 0u);
    cur = nxt;
    nxt += nn;
    n = nn;
  }
  HIP_CHECK(hipGetLastError());
  if (lc.sib.size() > 32) throw std::runtime_error("blake3_late_chunk0: too many levels");
}
// late half: chunk 0 (prefix with the commitment patched in from device memory, then the stream's words) and the leftmost path
void blake3_late_chunk0_finish(Ctx& ctx, const uint8_t* d_prefix, size_t prefix_len, const u64* d_words, size_t nwords, const Digest* d_cap,
                               size_t cap_off, size_t ncap, const LateChunk0& lc, size_t nchunks, Digest* out_dev, ChallengeBG* d_bg,
                               u32* d_state12) {
  if (prefix_len > 1024 || cap_off + 32 * ncap > prefix_len) throw std::runtime_error("blake3_late_chunk0: the prefix must lie inside chunk 0");
  ClaimsRootArgs a;
  memset(&a, 0, sizeof(a));
  a.prefix = d_prefix;
  a.words = d_words;
  a.pl = prefix_len;
  a.len = prefix_len + 8 * nwords;
  a.cap = reinterpret_cast<const u32*>(d_cap);
  a.cap_off = (u32)cap_off;
  a.cap_bytes = (u32)(32 * ncap);
  unsigned levels = 0;
  while ((size_t(1) << levels) < nchunks) levels++;
  if (lc.sib.size() != levels) throw std::runtime_error("blake3_late_chunk0: level count mismatch");
  for (unsigned l = 0; l < levels; l++) a.sib[l] = lc.sib[l];
  a.nchunks = (u32)nchunks;
  a.out = out_dev;
  a.bg = d_bg;
  a.state_out = d_state12;
  hipLaunchKernelGGL(claims_root_k, dim3(1), dim3(64), 0, ctx.stream, a);
  HIP_CHECK(hipGetLastError());
}

Digest blake3_device(Ctx& ctx, const uint8_t* d_prefix, size_t prefix_len, const u64* d_words, size_t nwords) {
  const size_t nchunks = blake3_num_chunks(prefix_len, nwords);
  DBuf<Digest> a(ctx, nchunks);
  blake3_chunk_cvs(ctx, d_prefix, prefix_len, d_words, nwords, 0, nchunks, a.p);
  return blake3_from_cvs(ctx, a.p, nchunks);
}

}  // namespace msamd
