// Per-row constraint / quotient evaluation on the quotient domain.
// Replaces quotient_values + quotient_values_inner (/root/reference/src/prover.rs:756-962), the node sweep
// ConstraintGraph::sweep_range (src/eval.rs:67-106) and logup_constraint_values (src/lookup.rs:152-208, D = 2).
//
// One thread = one storage row t of the committed LDEs (natural quotient-domain index i = bitrev(t)), so the
// "current row" loads are perfectly coalesced and the "next row" (i + q) maps to another contiguous run. The
// compiled node vector is lowered on the host into a register-allocated straight-line program (slots reused
// after a node's last use); slots live in LDS as [slot][lane] (bank-conflict free) or, for very large
// circuits, in a global scratch with the same layout. Selectors are computed from x = 7 w^i in-kernel.
#include <algorithm>

#include "msamd.h"
#include "program.h"
#include "quotient_params.h"

namespace msamd {

namespace {

static_assert(TW_LOG == 28 && TW_HALF == 14, "quotient_params.h repeats these for the hiprtc build");

template <bool LDS>
__global__ __launch_bounds__(256) void quotient_k(QParams p) {
  extern __shared__ u64 sm[];
  const size_t lt = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (lt >= p.rows) return;  // no barriers below
  const size_t t = p.row0 + lt;
  const unsigned lognq = p.log_n + p.log_q;
  const size_t nq = size_t(1) << lognq;
  const u32 i = bitrev32((u32)t, lognq);
  const u32 inext = (i + (1u << p.log_q)) & (u32)(nq - 1);
  const size_t tn = bitrev32(inext, lognq);
  u64* slots = LDS ? (sm + threadIdx.x) : (p.scratch + lt);
  const size_t stride = LDS ? blockDim.x : p.rows;

  // x = 7 * w_{nq}^i and the Lagrange selectors (p3 selectors_on_coset, unnormalised)
  u64 x = gl_mul_small(gl_mul(p.t1[(i << (TW_LOG - lognq)) >> TW_HALF], p.t0[(i << (TW_LOG - lognq)) & ((1u << TW_HALF) - 1)]), 7);
  const u32 qi = i & ((1u << p.log_q) - 1);
  const u64 zh = p.zh[qi];
  u64 d_first = gl_sub(x, 1), d_last = gl_sub(x, p.g_inv);
  u64 inv_both = gl_inv(gl_mul(d_first, d_last));
  u64 is_first = gl_mul(zh, gl_mul(inv_both, d_last));
  u64 is_last = gl_mul(zh, gl_mul(inv_both, d_first));
  u64 is_trans = d_last;

  for (u32 pc = 0; pc < p.n_instr; pc++) {
    const uint4 ins = reinterpret_cast<const uint4*>(p.code)[pc];
    u64 v;
    switch (ins.x) {
      case OP_CONST: v = p.consts[ins.z]; break;
      case OP_VAR: {
        u32 src = ins.z & 0xff, off = ins.z >> 8;
        size_t row = off ? tn : t;
        if (src == 1)
          v = p.s1[size_t(ins.w) * p.s1_h + row];
        else if (src == 0)
          v = p.pre[size_t(ins.w) * p.pre_h + row];
        else
          v = p.s2[size_t(ins.w) * p.s2_h + row];
        break;
      }
      case OP_PUBLIC: v = p.dyn->publics[ins.z]; break;
      case OP_IS_FIRST: v = is_first; break;
      case OP_IS_LAST: v = is_last; break;
      case OP_IS_TRANS: v = is_trans; break;
      case OP_ADD: v = gl_add(slots[ins.z * stride], slots[ins.w * stride]); break;
      case OP_SUB: v = gl_sub(slots[ins.z * stride], slots[ins.w * stride]); break;
      case OP_MUL: v = gl_mul(slots[ins.z * stride], slots[ins.w * stride]); break;
      default: v = gl_neg(slots[ins.z * stride]); break;
    }
    slots[ins.y * stride] = v;
  }

  // fold: user roots first, then the logUp values, constraint i weighted by alpha^{k-1-i}
  GlAcc fa0, fa1;  // unreduced sums of (constraint value) x (alpha power coordinate)
  acc_init(fa0);
  acc_init(fa1);
  u32 ci = 0;
  for (u32 z = 0; z < p.n_zeros; z++, ci++) {
    u64 cv = slots[p.zero_slots[z] * stride];
    E2 a = p.alpha_rev[ci];
    acc_mad(fa0, cv, a.c0);
    acc_mad(fa1, cv, a.c1);
  }
  const u64 beta0 = p.dyn->publics[0], beta1 = p.dyn->publics[1], gamma0 = p.dyn->publics[2], gamma1 = p.dyn->publics[3];
  const u64 inj0 = gl_mul(is_last, p.dyn->delta_scaled[0]), inj1 = gl_mul(is_last, p.dyn->delta_scaled[1]);
  auto fold2 = [&](u64 c0, u64 c1) {
    E2 a = p.alpha_rev[ci], b = p.alpha_rev[ci + 1];
    acc_mad(fa0, c0, a.c0);
    acc_mad(fa0, c1, b.c0);
    acc_mad(fa1, c0, a.c1);
    acc_mad(fa1, c1, b.c1);
    ci += 2;
  };
  if (p.n_lookups == 0) {
    u64 c0 = gl_add(gl_sub(p.s2[tn], p.s2[t]), inj0);
    u64 c1 = gl_add(gl_sub(p.s2[p.s2_h + tn], p.s2[p.s2_h + t]), inj1);
    fold2(c0, c1);
  } else {
    const uint32_t* ls = p.lookup_slots;
    u64 src0 = p.s2[t], src1 = p.s2[p.s2_h + t];
    for (u32 j = 0; j < p.n_lookups; j++) {
      u32 mslot = ls[0], na = ls[1];
      u64 tgt0, tgt1;
      if (j + 1 < p.n_lookups) {
        tgt0 = p.s2[size_t(2 * j + 2) * p.s2_h + t];
        tgt1 = p.s2[size_t(2 * j + 3) * p.s2_h + t];
      } else {
        tgt0 = gl_add(p.s2[tn], inj0);
        tgt1 = gl_add(p.s2[p.s2_h + tn], inj1);
      }
      // fingerprint = sum_k args[k] gamma^k (src/lookup.rs:192-197)
      u64 f0 = 0, f1 = 0;
      if (na <= 32) {
        GlAcc g0, g1;
        acc_init(g0);
        acc_init(g1);
        for (u32 k = 0; k < na; k++) {
          const u64 v = slots[ls[2 + k] * stride];
          acc_mad(g0, v, p.dyn->gpow[k].c0);
          acc_mad(g1, v, p.dyn->gpow[k].c1);
        }
        f0 = acc_reduce(g0);
        f1 = acc_reduce(g1);
      } else {
        for (u32 k = na; k-- > 0;) {
          u64 g0, g1;
          mul2(f0, f1, gamma0, gamma1, g0, g1);
          f0 = gl_add(g0, slots[ls[2 + k] * stride]);
          f1 = g1;
        }
      }
      u64 c0, c1;
      mul2(gl_add(f0, beta0), gl_add(f1, beta1), gl_sub(tgt0, src0), gl_sub(tgt1, src1), c0, c1);
      fold2(gl_sub(c0, slots[mslot * stride]), c1);
      if (j + 1 < p.n_lookups) {
        src0 = tgt0;
        src1 = tgt1;
      }
      ls += 2 + na;
    }
  }
  const u64 iv = p.zh_inv[qi];
  p.out[t] = gl_mul(acc_reduce(fa0), iv);
  p.out[nq + t] = gl_mul(acc_reduce(fa1), iv);
}


// ---- one WAVE per row, for short circuits with large programs (the reference's BLAKE3 compression circuit: 512 rows, 6952
// nodes, 73 lookups). A thread per row walks such a program alone - thousands of dependent steps at memory or LDS latency
// while the chip idles; here the program is scheduled by dependency level (build_program), the 64 lanes evaluate 64 nodes
// of one level per step, the row's slot file (one slot per position) lives in LDS, and the fold is split over the lanes:
// constraint roots and lookups are dealt out lane by lane, each lane keeps partial sums and lane 0 adds the 64 partials.
struct WaveArgs {
  const uint4* code;
  const uint32_t* zero_pos;
  const uint32_t* lookups;
  const uint32_t* lookup_off;
  uint32_t n_steps, n_leaf_steps;
};
enum : uint32_t { OP_NOP = 15 };

__global__ __launch_bounds__(64) void quotient_wave_k(QParams p, WaveArgs w) {
  extern __shared__ u64 sm[];
  const u32 lane = threadIdx.x;
  const size_t t = p.row0 + blockIdx.x;
  const unsigned lognq = p.log_n + p.log_q;
  const size_t nq = size_t(1) << lognq;
  const u32 i = bitrev32((u32)t, lognq);
  const u32 inext = (i + (1u << p.log_q)) & (u32)(nq - 1);
  const size_t tn = bitrev32(inext, lognq);
  u64 x = gl_mul_small(gl_mul(p.t1[(i << (TW_LOG - lognq)) >> TW_HALF], p.t0[(i << (TW_LOG - lognq)) & ((1u << TW_HALF) - 1)]), 7);
  const u32 qi = i & ((1u << p.log_q) - 1);
  const u64 zh = p.zh[qi];
  u64 d_first = gl_sub(x, 1), d_last = gl_sub(x, p.g_inv);
  u64 inv_both = gl_inv(gl_mul(d_first, d_last));
  const u64 is_first = gl_mul(zh, gl_mul(inv_both, d_last));
  const u64 is_last = gl_mul(zh, gl_mul(inv_both, d_first));
  const u64 is_trans = d_last;

  auto leaf = [&](const uint4 ins) -> u64 {
    switch (ins.x) {
      case OP_CONST: return p.consts[ins.z];
      case OP_VAR: {
        const u32 src = ins.z & 0xff, off = ins.z >> 8;
        const size_t row = off ? tn : t;
        if (src == 1) return p.s1[size_t(ins.w) * p.s1_h + row];
        if (src == 0) return p.pre[size_t(ins.w) * p.pre_h + row];
        return p.s2[size_t(ins.w) * p.s2_h + row];
      }
      case OP_PUBLIC: return p.dyn->publics[ins.z];
      case OP_IS_FIRST: return is_first;
      case OP_IS_LAST: return is_last;
      case OP_IS_TRANS: return is_trans;
      default: return 0;
    }
  };
  // level 0 (columns, constants, selectors): no dependencies, four steps' loads in flight
  u32 s = 0;
  for (; s + 4 <= w.n_leaf_steps; s += 4) {
    uint4 ins[4];
    u64 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) ins[k] = w.code[(s + k) * 64 + lane];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = leaf(ins[k]);
#pragma unroll
    for (int k = 0; k < 4; k++) sm[(s + k) * 64 + lane] = v[k];
  }
  for (; s < w.n_leaf_steps; s++) sm[s * 64 + lane] = leaf(w.code[s * 64 + lane]);
  __syncthreads();
  uint4 nxt = s < w.n_steps ? w.code[s * 64 + lane] : make_uint4(OP_NOP, 0, 0, 0);
  for (; s < w.n_steps; s++) {
    const uint4 ins = nxt;
    if (s + 1 < w.n_steps) nxt = w.code[(s + 1) * 64 + lane];  // the next step's instruction travels while this one runs
    u64 v = 0;
    switch (ins.x) {
      case OP_ADD: v = gl_add(sm[ins.z], sm[ins.w]); break;
      case OP_SUB: v = gl_sub(sm[ins.z], sm[ins.w]); break;
      case OP_MUL: v = gl_mul(sm[ins.z], sm[ins.w]); break;
      case OP_NEG: v = gl_neg(sm[ins.z]); break;
      case OP_NOP: break;
      default: v = leaf(ins); break;  // (leaves sit in level 0; kept for completeness)
    }
    sm[s * 64 + lane] = v;
    if (ins.y) __syncthreads();  // the last step of a level (the flag is the same in all 64 lanes): the next level reads it
  }
  __syncthreads();

  // fold: constraint i weighted by alpha^{k-1-i}; user roots first, then two values per lookup (or the plain pair)
  GlAcc fa0, fa1;
  acc_init(fa0);
  acc_init(fa1);
  for (u32 z = lane; z < p.n_zeros; z += 64) {
    const u64 cv = sm[w.zero_pos[z]];
    const E2 a = p.alpha_rev[z];
    acc_mad(fa0, cv, a.c0);
    acc_mad(fa1, cv, a.c1);
  }
  const u64 beta0 = p.dyn->publics[0], beta1 = p.dyn->publics[1], gamma0 = p.dyn->publics[2], gamma1 = p.dyn->publics[3];
  const u64 inj0 = gl_mul(is_last, p.dyn->delta_scaled[0]), inj1 = gl_mul(is_last, p.dyn->delta_scaled[1]);
  auto fold2 = [&](u32 ci, u64 c0, u64 c1) {
    const E2 a = p.alpha_rev[ci], b = p.alpha_rev[ci + 1];
    acc_mad(fa0, c0, a.c0);
    acc_mad(fa0, c1, b.c0);
    acc_mad(fa1, c0, a.c1);
    acc_mad(fa1, c1, b.c1);
  };
  if (p.n_lookups == 0) {
    if (lane == 0) {
      const u64 c0 = gl_add(gl_sub(p.s2[tn], p.s2[t]), inj0);
      const u64 c1 = gl_add(gl_sub(p.s2[p.s2_h + tn], p.s2[p.s2_h + t]), inj1);
      fold2(p.n_zeros, c0, c1);
    }
  } else {
    for (u32 j = lane; j < p.n_lookups; j += 64) {
      const uint32_t* ls = w.lookups + w.lookup_off[j];
      const u32 mpos = ls[0], na = ls[1];
      const u64 src0 = p.s2[size_t(2 * j) * p.s2_h + t], src1 = p.s2[size_t(2 * j + 1) * p.s2_h + t];
      u64 tgt0, tgt1;
      if (j + 1 < p.n_lookups) {
        tgt0 = p.s2[size_t(2 * j + 2) * p.s2_h + t];
        tgt1 = p.s2[size_t(2 * j + 3) * p.s2_h + t];
      } else {
        tgt0 = gl_add(p.s2[tn], inj0);
        tgt1 = gl_add(p.s2[p.s2_h + tn], inj1);
      }
      u64 f0 = 0, f1 = 0;  // fingerprint = sum_k args[k] gamma^k (src/lookup.rs:192-197)
      if (na <= 32) {
        GlAcc g0, g1;
        acc_init(g0);
        acc_init(g1);
        for (u32 k = 0; k < na; k++) {
          const u64 v = sm[ls[2 + k]];
          acc_mad(g0, v, p.dyn->gpow[k].c0);
          acc_mad(g1, v, p.dyn->gpow[k].c1);
        }
        f0 = acc_reduce(g0);
        f1 = acc_reduce(g1);
      } else {
        for (u32 k = na; k-- > 0;) {
          u64 g0, g1;
          mul2(f0, f1, gamma0, gamma1, g0, g1);
          f0 = gl_add(g0, sm[ls[2 + k]]);
          f1 = g1;
        }
      }
      u64 c0, c1;
      mul2(gl_add(f0, beta0), gl_add(f1, beta1), gl_sub(tgt0, src0), gl_sub(tgt1, src1), c0, c1);
      fold2(p.n_zeros + 2 * j, gl_sub(c0, sm[mpos]), c1);
    }
  }
  __syncthreads();  // every lane is done with the slot file: its first 128 words now carry the lanes' partial sums
  sm[lane] = acc_reduce(fa0);
  sm[64 + lane] = acc_reduce(fa1);
  __syncthreads();
  if (lane == 0) {
    u64 s0 = 0, s1 = 0;
    for (u32 k = 0; k < 64; k++) {
      s0 = gl_add(s0, sm[k]);
      s1 = gl_add(s1, sm[64 + k]);
    }
    const u64 iv = p.zh_inv[qi];
    p.out[t] = gl_mul(s0, iv);
    p.out[nq + t] = gl_mul(s1, iv);
  }
}


// ---- SystemWitness::from_stage_1 on the device (src/system.rs:275-328): the lookup prefix of the node program is
// swept over every trace row (with wrap-around for the next-row window) and the multiplicity / argument values are
// written into the flat LookupValues storage (src/lookup.rs:392-405). Traces are row-major here, as uploaded.
struct LvParams {
  const u64* trace;   // h x main_w row-major
  const u64* pre;     // h x pre_w row-major (or null)
  size_t h;
  uint32_t main_w, pre_w;
  const uint32_t* code;
  const u64* consts;
  const uint32_t* lookup_slots;
  uint32_t n_instr, n_lookups, n_slots, args_w;
  u64* mult;          // h x L
  u64* args;          // h x args_w
};
__global__ __launch_bounds__(256) void lookup_values_k(LvParams p) {
  extern __shared__ u64 sm[];
  const size_t r = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
  if (r >= p.h) return;
  const size_t rn = r + 1 == p.h ? 0 : r + 1;
  u64* slots = sm + threadIdx.x;
  const size_t stride = blockDim.x;
  const u64 is_first = r == 0, is_last = r + 1 == p.h, is_trans = r + 1 != p.h;
  for (u32 pc = 0; pc < p.n_instr; pc++) {
    const uint4 ins = reinterpret_cast<const uint4*>(p.code)[pc];
    u64 v;
    switch (ins.x) {
      case OP_CONST: v = p.consts[ins.z]; break;
      case OP_VAR: {
        u32 src = ins.z & 0xff, off = ins.z >> 8;
        size_t row = off ? rn : r;
        v = src == 1 ? p.trace[row * p.main_w + ins.w] : p.pre[row * p.pre_w + ins.w];
        break;
      }
      case OP_IS_FIRST: v = is_first; break;
      case OP_IS_LAST: v = is_last; break;
      case OP_IS_TRANS: v = is_trans; break;
      case OP_ADD: v = gl_add(slots[ins.z * stride], slots[ins.w * stride]); break;
      case OP_SUB: v = gl_sub(slots[ins.z * stride], slots[ins.w * stride]); break;
      case OP_MUL: v = gl_mul(slots[ins.z * stride], slots[ins.w * stride]); break;
      case OP_NEG: v = gl_neg(slots[ins.z * stride]); break;
      default: v = 0; break;  // publics / stage-2 columns cannot occur in lookup expressions (checked on the host)
    }
    slots[ins.y * stride] = v;
  }
  const uint32_t* ls = p.lookup_slots;
  u64* arow = p.args + r * p.args_w;
  for (u32 j = 0; j < p.n_lookups; j++) {
    const u32 mslot = ls[0], na = ls[1];
    p.mult[r * p.n_lookups + j] = slots[mslot * stride];
    for (u32 k = 0; k < na; k++) arow[k] = slots[ls[2 + k] * stride];
    arow += na;
    ls += 2 + na;
  }
}

}  // namespace

// ---- host: lower the compiled node vector to a slot-allocated program
void build_program(Ctx& ctx, const std::vector<PNode>& nodes, const std::vector<uint32_t>& zeros,
                   const std::vector<std::pair<uint32_t, std::vector<uint32_t>>>& lookups, DProgram& out) {
  const size_t nn = nodes.size();
  const uint32_t INF = 0xFFFFFFFFu;
  std::vector<uint32_t> last_use(nn, 0);
  std::vector<char> needed(nn, 0);
  for (auto z : zeros) {
    needed[z] = 1;
    last_use[z] = INF;
  }
  for (auto& l : lookups) {
    needed[l.first] = 1;
    last_use[l.first] = INF;
    for (auto a : l.second) {
      needed[a] = 1;
      last_use[a] = INF;
    }
  }
  for (size_t i = nn; i-- > 0;) {
    if (!needed[i]) continue;
    const PNode& n = nodes[i];
    auto use = [&](uint64_t c) {
      needed[c] = 1;
      if (last_use[c] != INF && last_use[c] < i) last_use[c] = (uint32_t)i;
    };
    if (n.kind == OP_ADD || n.kind == OP_SUB || n.kind == OP_MUL) {
      use(n.a);
      use(n.b);
    } else if (n.kind == OP_NEG) {
      use(n.a);
    }
  }
  std::vector<uint32_t> slot_of(nn, INF), free_slots, code;
  std::vector<u64> consts;
  uint32_t n_slots = 0;
  for (size_t i = 0; i < nn; i++) {
    if (!needed[i]) continue;
    const PNode& n = nodes[i];
    uint32_t a = 0, b = 0;
    switch (n.kind) {
      case OP_CONST:
        a = (uint32_t)consts.size();
        consts.push_back(n.a);
        break;
      case OP_VAR:
        a = n.source | (n.offset << 8);
        b = (uint32_t)n.a;
        break;
      case OP_PUBLIC: a = (uint32_t)n.a; break;
      case OP_ADD:
      case OP_SUB:
      case OP_MUL:
        a = slot_of[n.a];
        b = slot_of[n.b];
        break;
      case OP_NEG: a = slot_of[n.a]; break;
      default: break;
    }
    // operands whose last use is this node free their slots (the destination may reuse them: reads precede the write)
    auto maybe_free = [&](uint64_t c) {
      if (last_use[c] == i && slot_of[c] != INF) {
        free_slots.push_back(slot_of[c]);
        slot_of[c] = INF;
      }
    };
    if (n.kind == OP_ADD || n.kind == OP_SUB || n.kind == OP_MUL) {
      maybe_free(n.a);
      if (n.b != n.a) maybe_free(n.b);
    } else if (n.kind == OP_NEG) {
      maybe_free(n.a);
    }
    uint32_t dst;
    if (!free_slots.empty()) {
      dst = free_slots.back();
      free_slots.pop_back();
    } else {
      dst = n_slots++;
    }
    slot_of[i] = dst;
    code.push_back(n.kind);
    code.push_back(dst);
    code.push_back(a);
    code.push_back(b);
    if (last_use[i] == 0 && !(last_use[i] == INF)) {
      // never used (can only happen for a root-less needed node): keep the slot, harmless
    }
  }
  if (n_slots == 0) n_slots = 1;
  std::vector<uint32_t> zslots, lslots;
  for (auto z : zeros) zslots.push_back(slot_of[z]);
  for (auto& l : lookups) {
    lslots.push_back(slot_of[l.first]);
    lslots.push_back((uint32_t)l.second.size());
    for (auto a : l.second) lslots.push_back(slot_of[a]);
  }
  // ---- the wave schedule (quotient_wave_k): nodes ordered by dependency level, 64 per step, a level padded to whole steps
  {
    size_t n_needed = 0;
    for (size_t i = 0; i < nn; i++) n_needed += needed[i] != 0;
    out.wave_steps = out.wave_leaf_steps = 0;
    if (n_needed >= 1024) {
      std::vector<uint32_t> level(nn, 0);
      uint32_t max_level = 0;
      for (size_t i = 0; i < nn; i++) {
        if (!needed[i]) continue;
        const PNode& n = nodes[i];
        if (n.kind == OP_ADD || n.kind == OP_SUB || n.kind == OP_MUL)
          level[i] = 1 + std::max(level[n.a], level[n.b]);
        else if (n.kind == OP_NEG)
          level[i] = 1 + level[n.a];
        max_level = std::max(max_level, level[i]);
      }
      std::vector<std::vector<uint32_t>> by_level(max_level + 1);
      for (size_t i = 0; i < nn; i++)
        if (needed[i]) by_level[level[i]].push_back((uint32_t)i);
      std::vector<uint32_t> pos_of(nn, INF), wcode;  // (the wave program's constants are appended to `consts`)
      size_t pos = 0;
      for (uint32_t lv = 0; lv <= max_level; lv++) {
        for (uint32_t i : by_level[lv]) pos_of[i] = (uint32_t)pos++;
        pos = (pos + 63) & ~size_t(63);
        if (lv == 0) out.wave_leaf_steps = pos / 64;
      }
      const size_t n_pos = pos;
      if (n_pos * 8 <= 160 * 1024) {
        wcode.assign(n_pos * 4, 0);
        for (size_t q = 0; q < n_pos; q++) wcode[4 * q] = 15;  // OP_NOP
        size_t at = 0;
        for (uint32_t lv = 0; lv <= max_level; lv++) {
          for (uint32_t i : by_level[lv]) {
            const PNode& n = nodes[i];
            uint32_t a = 0, b = 0;
            switch (n.kind) {
              case OP_CONST:
                a = (uint32_t)consts.size();
                consts.push_back(n.a);
                break;
              case OP_VAR:
                a = n.source | (n.offset << 8);
                b = (uint32_t)n.a;
                break;
              case OP_PUBLIC: a = (uint32_t)n.a; break;
              case OP_ADD:
              case OP_SUB:
              case OP_MUL:
                a = pos_of[n.a];
                b = pos_of[n.b];
                break;
              case OP_NEG: a = pos_of[n.a]; break;
              default: break;
            }
            uint32_t* w4 = &wcode[4 * pos_of[i]];
            w4[0] = n.kind, w4[2] = a, w4[3] = b;
            at = pos_of[i] + 1;
          }
          at = (at + 63) & ~size_t(63);
          if (at >= 64)
            for (size_t q = at - 64; q < at; q++) wcode[4 * q + 1] = 1;  // the level's last step: a barrier follows it
        }
        std::vector<uint32_t> zpos, lk, lk_off;
        for (auto z : zeros) zpos.push_back(pos_of[z]);
        for (auto& l : lookups) {
          lk_off.push_back((uint32_t)lk.size());
          lk.push_back(pos_of[l.first]);
          lk.push_back((uint32_t)l.second.size());
          for (auto a : l.second) lk.push_back(pos_of[a]);
        }
        out.wave_steps = n_pos / 64;
        out.wave_code = DBuf<uint32_t>(ctx, wcode.size());
        out.wave_zero_pos = DBuf<uint32_t>(ctx, std::max<size_t>(zpos.size(), 1));
        out.wave_lookups = DBuf<uint32_t>(ctx, std::max<size_t>(lk.size(), 1));
        out.wave_lookup_off = DBuf<uint32_t>(ctx, std::max<size_t>(lk_off.size(), 1));
        ctx.h2d(out.wave_code.p, wcode.data(), wcode.size() * 4);
        if (!zpos.empty()) ctx.h2d(out.wave_zero_pos.p, zpos.data(), zpos.size() * 4);
        if (!lk.empty()) ctx.h2d(out.wave_lookups.p, lk.data(), lk.size() * 4);
        if (!lk_off.empty()) ctx.h2d(out.wave_lookup_off.p, lk_off.data(), lk_off.size() * 4);
        ctx.sync();  // (the vectors above go out of scope)
      } else {
        out.wave_leaf_steps = 0;
      }
    }
  }
  out.n_instr = code.size() / 4;
  out.n_slots = n_slots;
  out.n_zeros = zeros.size();
  out.n_lookups = lookups.size();
  out.code = DBuf<uint32_t>(ctx, std::max<size_t>(code.size(), 4));
  out.consts = DBuf<u64>(ctx, std::max<size_t>(consts.size(), 1));
  out.zero_slots = DBuf<uint32_t>(ctx, std::max<size_t>(zslots.size(), 1));
  out.lookup_slots = DBuf<uint32_t>(ctx, std::max<size_t>(lslots.size(), 1));
  if (!code.empty()) ctx.h2d(out.code.p, code.data(), code.size() * 4);
  if (!consts.empty()) ctx.h2d(out.consts.p, consts.data(), consts.size() * 8);
  if (!zslots.empty()) ctx.h2d(out.zero_slots.p, zslots.data(), zslots.size() * 4);
  if (!lslots.empty()) ctx.h2d(out.lookup_slots.p, lslots.data(), lslots.size() * 4);
  ctx.sync();
}


bool lookup_values_device(Ctx& ctx, const DProgram& prefix, const u64* d_trace, const u64* d_pre, size_t h, size_t main_w,
                          size_t pre_w, size_t args_w, u64* d_mult, u64* d_args, hipStream_t on_stream) {
  unsigned threads = 256;
  while (threads > 64 && prefix.n_slots * threads * 8 > 64 * 1024) threads >>= 1;
  if (prefix.n_slots * threads * 8 > 64 * 1024) return false;  // very large prefix: the caller sweeps on the host
  LvParams p;
  p.trace = d_trace;
  p.pre = d_pre;
  p.h = h;
  p.main_w = (uint32_t)main_w;
  p.pre_w = (uint32_t)pre_w;
  p.code = prefix.code.p;
  p.consts = prefix.consts.p;
  p.lookup_slots = prefix.lookup_slots.p;
  p.n_instr = (uint32_t)prefix.n_instr;
  p.n_lookups = (uint32_t)prefix.n_lookups;
  p.n_slots = (uint32_t)prefix.n_slots;
  p.args_w = (uint32_t)args_w;
  p.mult = d_mult;
  p.args = d_args;
  hipLaunchKernelGGL(lookup_values_k, dim3((unsigned)((h + threads - 1) / threads)), dim3(threads), prefix.n_slots * threads * 8,
                     on_stream ? on_stream : ctx.stream, p);
  HIP_CHECK(hipGetLastError());
  return true;
}

u64 quotient_inj_norm(unsigned log_n) {  // 1 / (n g), src/prover.rs:782-784
  return gl_inv(gl_mul((u64)(size_t(1) << log_n) % GL_P, gl_two_adic_generator(log_n)));
}

// lanes per workgroup for a short circuit whose slot file does not fit the usual 64 KB at 64 lanes (0 = use the global scratch):
// the largest of 32 .. 4 whose slots fit the LDS a workgroup may have (160 KB on gfx950, opted into once), as long as the rows
// make at most a few workgroups per CU - a tall circuit walks the global scratch at full width instead
// Dynamic LDS above 64 KB has to be opted into per kernel AND per device (a process may drive several GPUs, one context each:
// the in-process transport's thread ranks), so the attribute is set on the current device in front of every such launch -
// these are the rare paths of large programs, the call costs microseconds.
static size_t big_lds_limit(const void* kernel) {
  const size_t want = 160 * 1024;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want) == hipSuccess) return want;
  (void)hipGetLastError();
  return size_t(64) * 1024;
}
static bool wave_lds_ok(size_t bytes) { return bytes <= big_lds_limit(reinterpret_cast<const void*>(&quotient_wave_k)); }
static unsigned few_lanes_per_workgroup(size_t n_slots, size_t nq) {
  if (n_slots * 4 * 8 > 160 * 1024 || (nq + 31) / 32 > 256) return 0;  // cannot apply whatever the limit is
  const size_t lds_max = big_lds_limit(reinterpret_cast<const void*>(&quotient_k<true>));
  if (getenv("MSAMD_NO_FEW_LANES")) return 0;
  for (unsigned th = 32; th >= 4; th >>= 1)
    if (n_slots * th * 8 <= lds_max && (nq + th - 1) / th <= 256) return th;  // one round of workgroups over the CUs
  return 0;
}

void quotient_eval(Ctx& ctx, const DProgram& prog, const QuotientArgs& a, u64* out) {
  const unsigned lognq = a.log_n + a.log_q;
  if (lognq > 26) throw std::runtime_error("quotient domain larger than 2^26 is not supported");
  const size_t n = size_t(1) << a.log_n, q = size_t(1) << a.log_q, nq = n * q;
  // Z_H on the coset: x^n - 1 = 7^n w_q^j - 1, period q (src/prover.rs:775 -> p3 selectors_on_coset)
  std::vector<u64> zh(q), zhi(q);
  u64 s_pow_n = gl_exp_pow2(GL_GEN, a.log_n), wq = gl_two_adic_generator(a.log_q), xx = 1;
  for (size_t j = 0; j < q; j++) {
    zh[j] = gl_sub(gl_mul(s_pow_n, xx), 1);
    zhi[j] = gl_inv(zh[j]);
    xx = gl_mul(xx, wq);
  }
  const size_t k = prog.constraint_count;
  const bool inl = prog.jit.function && prog.jit.inline_tables && q <= 8 && k <= QP_INLINE_ALPHA;  // Z_H tables in the argument block
  DBuf<u64> dzh;
  if (!inl) {
    dzh = DBuf<u64>(ctx, 2 * q);
    ctx.h2d(dzh.p, zh.data(), q * 8);
    ctx.h2d(dzh.p + q, zhi.data(), q * 8);
  }
  const u64 g = gl_two_adic_generator(a.log_n);
  // the challenge-dependent block: given by the caller in device memory (device transcript), or built here from the values
  DBuf<uint8_t> dyn_buf;
  const QDyn* dyn = a.dyn;
  const E2* alpha_rev = a.alpha_rev;
  if (!dyn) {
    std::vector<uint8_t> host(sizeof(QDyn) + std::max<size_t>(k, 1) * sizeof(E2));
    QDyn* d = reinterpret_cast<QDyn*>(host.data());
    E2* arev = reinterpret_cast<E2*>(host.data() + sizeof(QDyn));
    quotient_dyn_fill(*d, arev, k, a.publics, a.alpha, quotient_inj_norm(a.log_n));
    dyn_buf = DBuf<uint8_t>(ctx, host.size());
    ctx.h2d(dyn_buf.p, host.data(), host.size());
    dyn = reinterpret_cast<const QDyn*>(dyn_buf.p);
    alpha_rev = reinterpret_cast<const E2*>(dyn_buf.p + sizeof(QDyn));
  }

  QParams p;
  p.pre = a.pre;
  p.s1 = a.s1;
  p.s2 = a.s2;
  p.pre_h = a.pre_h;
  p.s1_h = a.s1_h;
  p.s2_h = a.s2_h;
  p.log_n = a.log_n;
  p.log_q = a.log_q;
  p.dyn = dyn;
  p.g_inv = gl_inv(g);
  p.zh = dzh.p;
  p.zh_inv = dzh.p ? dzh.p + q : nullptr;
  p.alpha_rev = alpha_rev;
  memset(p.zh_in, 0, sizeof(p.zh_in));
  memset(p.zh_inv_in, 0, sizeof(p.zh_inv_in));
  if (inl) {
    for (size_t j = 0; j < 8; j++) {  // period q: any lane index masked with 7 lands on the right entry
      p.zh_in[j] = zh[j % q];
      p.zh_inv_in[j] = zhi[j % q];
    }
  }
  p.code = prog.code.p;
  p.consts = prog.consts.p;
  p.zero_slots = prog.zero_slots.p;
  p.lookup_slots = prog.lookup_slots.p;
  p.n_instr = (uint32_t)prog.n_instr;
  p.n_zeros = (uint32_t)prog.n_zeros;
  p.n_lookups = (uint32_t)prog.n_lookups;
  p.n_slots = (uint32_t)prog.n_slots;
  p.t0 = ctx.tw0;
  p.t1 = ctx.tw1;
  p.out = out;
  p.scratch = nullptr;

  const double bytes = double(nq) * 8.0 * (2.0 * (prog.main_w + prog.s2_w + prog.pre_w) + 2.0);
  unsigned threads = 256;
  while (threads > 64 && prog.n_slots * threads * 8 > 64 * 1024) threads >>= 1;
  if (prog.jit.function) {
    p.row0 = 0;
    p.rows = nq;
    hipEvent_t ev = ctx.prof_begin(K_QUOTIENT);
    quotient_jit_launch(ctx, prog.jit, p, nq);
    ctx.prof_end(K_QUOTIENT, ev, bytes);
  } else if (prog.n_slots * threads * 8 <= 64 * 1024) {
    p.row0 = 0;
    p.rows = nq;
    hipEvent_t ev = ctx.prof_begin(K_QUOTIENT);
    hipLaunchKernelGGL(quotient_k<true>, dim3((unsigned)((nq + threads - 1) / threads)), dim3(threads), prog.n_slots * threads * 8,
                       ctx.stream, p);
    ctx.prof_end(K_QUOTIENT, ev, bytes);
  } else if (prog.wave_steps && nq <= 16384 && wave_lds_ok(prog.wave_steps * 64 * 8) && !getenv("MSAMD_NO_WAVE_QUOTIENT")) {
    // a short circuit with a large program: one wave per row over the level-scheduled program (quotient_wave_k)
    p.row0 = 0;
    p.rows = nq;
    WaveArgs w;
    w.code = reinterpret_cast<const uint4*>(prog.wave_code.p);
    w.zero_pos = prog.wave_zero_pos.p;
    w.lookups = prog.wave_lookups.p;
    w.lookup_off = prog.wave_lookup_off.p;
    w.n_steps = (uint32_t)prog.wave_steps;
    w.n_leaf_steps = (uint32_t)prog.wave_leaf_steps;
    hipEvent_t ev = ctx.prof_begin(K_QUOTIENT);
    hipLaunchKernelGGL(quotient_wave_k, dim3((unsigned)nq), dim3(64), prog.wave_steps * 64 * 8, ctx.stream, p, w);
    ctx.prof_end(K_QUOTIENT, ev, bytes);
  } else if (const unsigned few = few_lanes_per_workgroup(prog.n_slots, nq)) {
    // A SHORT circuit with thousands of live slots (the reference's 2625-column BLAKE3 compression circuit: 512 rows, 6952
    // nodes): with the slot file in global memory two workgroups walk the program at the latency of three global accesses per
    // node (4.1 ms) while 254 CUs idle. Few lanes per workgroup instead, each workgroup with up to 160 KB of LDS for its
    // lanes' slot files: the rows spread over the chip and a node costs LDS latency.
    p.row0 = 0;
    p.rows = nq;
    hipEvent_t ev = ctx.prof_begin(K_QUOTIENT);
    hipLaunchKernelGGL(quotient_k<true>, dim3((unsigned)((nq + few - 1) / few)), dim3(few), prog.n_slots * few * 8, ctx.stream, p);
    ctx.prof_end(K_QUOTIENT, ev, bytes);
  } else {
    // global scratch in batches of rows so that it stays below ~1 GiB
    size_t batch = (size_t(1) << 30) / (prog.n_slots * 8);
    batch = std::max<size_t>(256, batch & ~size_t(255));
    batch = std::min(batch, nq);
    DBuf<u64> scratch(ctx, batch * prog.n_slots);
    p.scratch = scratch.p;
    for (size_t r0 = 0; r0 < nq; r0 += batch) {
      p.row0 = r0;
      p.rows = std::min(batch, nq - r0);
      hipEvent_t ev = ctx.prof_begin(K_QUOTIENT);
      hipLaunchKernelGGL(quotient_k<false>, dim3((unsigned)((p.rows + 255) / 256)), dim3(256), 0, ctx.stream, p);
      ctx.prof_end(K_QUOTIENT, ev, bytes * double(p.rows) / double(nq));
    }
  }
  HIP_CHECK(hipGetLastError());
}

}  // namespace msamd
