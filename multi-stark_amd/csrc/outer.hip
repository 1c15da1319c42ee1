// The OUTER Fiat-Shamir transcript on the device: beta and gamma (behind the claims), alpha (behind the stage-2 commitment
// and the accumulators) and zeta (behind the quotient commitment) are sampled by three one-workgroup kernels that sit in
// the stream between the commitments and the kernels that consume the challenges, so prove() queues stage 2, the quotient
// and the opened-value sums without a single host round trip in between (src/prover.rs:375-381, 425-433, 530-537; the
// challenger is SerializingChallenger64<Goldilocks, HashChallenger<u8, Blake3, 32>>, src/types.rs:28-29).
//
// As with the FRI commit phase (challenge_dev.h) the device is an accelerator, never the authority: the host challenger
// replays the same steps from the commitments once they have arrived (with the opened values) and prove() fails if any
// challenge differs.
//
// Each of the three transcript pieces is hashed as ONE BLAKE3 chunk (at most 1024 bytes): the challenger's pending input is
// the 32-byte digest left by the previous sample plus what was observed since. prove() keeps the host path when a piece
// would be longer (many circuits, a tall cap).
#include "outer_dev.h"
#include "quotient_params.h"

namespace msamd {
namespace {

__global__ __launch_bounds__(64) void outer_beta_gamma_k(const u32* __restrict__ digest, ChallengeBG* __restrict__ bg, u32* __restrict__ state_out) {
  outer_beta_gamma_step(digest, bg, state_out);
}

}  // namespace

struct OuterCircuit {
  QDyn* dyn;
  E2* alpha_rev;
  u64 inj_norm;  // 1 / (n g)
  u32 k;         // constraint count
  u32 pad;
};
struct OuterAlphaArgs {
  const u32* state_in;  // 12 words (outer_beta_gamma_k)
  const u32* cap;       // stage-2 commitment, 8 * ncap words
  const E2* tot;        // [0] = claims, [1 + pos] = circuit totals
  const ChallengeBG* bg;
  E2* accs;             // na + 1: the initial accumulator, then the running sums (src/prover.rs:382-409)
  E2* alpha_out;
  u32* state_out;       // 8 words
  const OuterCircuit* circuits;
  u32 ncap, na;
};

namespace {

constexpr u32 OUTER_MAX_NA = 59;  // 48 + 32 ncap + 16 na <= 1024

__global__ __launch_bounds__(256) void outer_alpha_k(OuterAlphaArgs a) {
  __shared__ u32 msg[256];
  __shared__ E2 sh_acc[OUTER_MAX_NA + 1];
  __shared__ E2 sh_alpha;
  const u32 t = threadIdx.x;
  if (t == 0) {
    E2 acc = a.tot[0];
    sh_acc[0] = acc;
    a.accs[0] = acc;
    for (u32 i = 0; i < a.na; i++) {
      acc = e2_add(acc, a.tot[1 + i]);
      sh_acc[1 + i] = acc;
      a.accs[1 + i] = acc;
    }
  }
  __syncthreads();
  const u32 cap_words = 8 * a.ncap, len_words = 12 + cap_words + 4 * a.na;
  {
    u32 v = 0;
    if (t < 12) v = a.state_in[t];
    else if (t < 12 + cap_words) v = a.cap[t - 12];
    else if (t < len_words) {
      const u32 j = t - 12 - cap_words;
      const E2 e = sh_acc[1 + j / 4];
      const u64 c = (j & 2) ? e.c1 : e.c0;
      v = (j & 1) ? (u32)(c >> 32) : (u32)c;
    }
    msg[t] = v;
  }
  __syncthreads();
  if (t == 0) {
    DevChallenger s;
    dc_hash_chunk(msg, 4 * len_words, s.dg);
    s.pos = 32;
    const E2 alpha = dc_sample_ext(s);
    *a.alpha_out = alpha;
    for (int k = 0; k < 8; k++) a.state_out[k] = s.dg[k];
    sh_alpha = alpha;
  }
  __syncthreads();
  const E2 alpha = sh_alpha;
  for (u32 c = 0; c < a.na; c++) {
    const OuterCircuit oc = a.circuits[c];
    if (t == 0) {
      const E2 four[4] = {a.bg->beta, a.bg->gamma, sh_acc[c], sh_acc[c + 1]};
      for (int k = 0; k < 4; k++) {
        oc.dyn->publics[2 * k] = four[k].c0;
        oc.dyn->publics[2 * k + 1] = four[k].c1;
      }
      oc.dyn->delta_scaled[0] = gl_mul(gl_sub(four[3].c0, four[2].c0), oc.inj_norm);
      oc.dyn->delta_scaled[1] = gl_mul(gl_sub(four[3].c1, four[2].c1), oc.inj_norm);
    }
    if (t >= 64 && t < 96) oc.dyn->gpow[t - 64] = a.bg->gp.g[t - 64];
    // reversed powers: each thread walks its own residue class with stride alpha^256
    if (t < oc.k) {
      E2 ap = e2_pow(alpha, t);
      const E2 step = e2_pow(alpha, 256);
      for (u32 i = t; i < oc.k; i += 256) {
        oc.alpha_rev[oc.k - 1 - i] = ap;
        ap = e2_mul(ap, step);
      }
    }
  }
}

// observe the quotient commitment, zeta <- sample; points[0] = zeta, points[1 + i] = zeta * g(2^lds[i]) (src/prover.rs:538-560)
__global__ __launch_bounds__(256) void outer_zeta_k(const u32* __restrict__ state_in, const u32* __restrict__ cap, u32 ncap,
                                                    const u32* __restrict__ lds, u32 n_ld, E2* __restrict__ points,
                                                    u32* __restrict__ state_out /* nullable: 8 words, the input buffer behind zeta */) {
  __shared__ u32 msg[256];
  __shared__ E2 sh_zeta;
  const u32 t = threadIdx.x;
  const u32 len_words = 8 + 8 * ncap;
  msg[t] = t < 8 ? state_in[t] : t < len_words ? cap[t - 8] : 0u;
  __syncthreads();
  if (t == 0) {
    DevChallenger s;
    dc_hash_chunk(msg, 4 * len_words, s.dg);
    s.pos = 32;
    const E2 zeta = dc_sample_ext(s);
    points[0] = zeta;
    sh_zeta = zeta;
    if (state_out)
      for (int k = 0; k < 8; k++) state_out[k] = s.dg[k];
  }
  __syncthreads();
  for (u32 i = t; i < n_ld; i += 256) points[1 + i] = e2_mul_base(sh_zeta, gl_two_adic_generator(lds[i]));
}

}  // namespace

namespace {
// A joint proof's logUp totals (prover_sharded.inc): every rank contributed a row [total of circuit 0 .. na - 1 | its share of
// the claims' sum] to one all_gather. tot[0] = the claims' sum over all ranks, tot[1 + pos] = circuit pos's total from the
// rank that computed it - the layout outer_alpha_k reads. Field sums are exact, so the order of the additions is immaterial.
struct JointTotalsArgs {
  const E2* all;  // world rows of na + 1 values
  E2* tot;        // na + 1
  u32 world, na;
  uint16_t src[OUTER_MAX_NA];  // which rank's row holds circuit pos's total
};
__global__ __launch_bounds__(64) void joint_totals_k(JointTotalsArgs a) {
  const u32 t = threadIdx.x;
  if (t < a.na) a.tot[1 + t] = a.all[(size_t)a.src[t] * (a.na + 1) + t];
  if (t == 63) {
    E2 acc = e2(0);
    for (u32 r = 0; r < a.world; r++) acc = e2_add(acc, a.all[(size_t)r * (a.na + 1) + a.na]);
    a.tot[0] = acc;
  }
}

}  // namespace

void outer_joint_totals(Ctx& ctx, const E2* d_all, size_t world, size_t na, const std::vector<int>& src_rank, E2* d_tot) {
  if (na < 1 || na > OUTER_MAX_NA || src_rank.size() != na || world > 65535) throw std::runtime_error("outer_joint_totals: shape");
  JointTotalsArgs a;
  memset(&a, 0, sizeof(a));
  a.all = d_all;
  a.tot = d_tot;
  a.world = (u32)world;
  a.na = (u32)na;
  for (size_t i = 0; i < na; i++) {
    if (src_rank[i] < 0 || (size_t)src_rank[i] >= world) throw std::runtime_error("outer_joint_totals: source rank out of range");
    a.src[i] = (uint16_t)src_rank[i];
  }
  hipLaunchKernelGGL(joint_totals_k, dim3(1), dim3(64), 0, ctx.stream, a);
  HIP_CHECK(hipGetLastError());
}

bool outer_fits(size_t ncap, size_t na) { return na >= 1 && na <= OUTER_MAX_NA && 48 + 32 * ncap + 16 * na <= 1024 && 32 + 32 * ncap <= 1024; }

void outer_beta_gamma(Ctx& ctx, const Digest* d_digest, ChallengeBG* d_bg, u32* d_state12) {
  hipLaunchKernelGGL(outer_beta_gamma_k, dim3(1), dim3(64), 0, ctx.stream, reinterpret_cast<const u32*>(d_digest), d_bg, d_state12);
  HIP_CHECK(hipGetLastError());
}

// circuits: per active circuit (in commitment order) where its QDyn block and reversed alpha powers go
void outer_alpha(Ctx& ctx, const u32* d_state12, const Digest* d_cap, size_t ncap, const E2* d_tot, size_t na, const ChallengeBG* d_bg,
                 const std::vector<OuterTarget>& targets, E2* d_accs, E2* d_alpha, u32* d_state8, DBuf<uint8_t>& keep) {
  if (!outer_fits(ncap, na) || targets.size() != na) throw std::runtime_error("outer_alpha: transcript piece does not fit one BLAKE3 chunk");
  std::vector<OuterCircuit> h(na);
  for (size_t i = 0; i < na; i++) {
    h[i].dyn = targets[i].dyn;
    h[i].alpha_rev = targets[i].alpha_rev;
    h[i].inj_norm = quotient_inj_norm(targets[i].log_n);
    h[i].k = (u32)targets[i].k;
    h[i].pad = 0;
  }
  keep = DBuf<uint8_t>(ctx, na * sizeof(OuterCircuit));
  ctx.h2d(keep.p, h.data(), na * sizeof(OuterCircuit));
  OuterAlphaArgs a;
  a.state_in = d_state12;
  a.cap = reinterpret_cast<const u32*>(d_cap);
  a.tot = d_tot;
  a.bg = d_bg;
  a.accs = d_accs;
  a.alpha_out = d_alpha;
  a.state_out = d_state8;
  a.circuits = reinterpret_cast<const OuterCircuit*>(keep.p);
  a.ncap = (u32)ncap;
  a.na = (u32)na;
  hipLaunchKernelGGL(outer_alpha_k, dim3(1), dim3(256), 0, ctx.stream, a);
  HIP_CHECK(hipGetLastError());
}

void outer_zeta(Ctx& ctx, const u32* d_state8, const Digest* d_cap, size_t ncap, const u32* d_lds, size_t n_ld, E2* d_points, u32* d_state_out) {
  if (32 + 32 * ncap > 1024) throw std::runtime_error("outer_zeta: transcript piece does not fit one BLAKE3 chunk");
  hipLaunchKernelGGL(outer_zeta_k, dim3(1), dim3(256), 0, ctx.stream, d_state8, reinterpret_cast<const u32*>(d_cap), (u32)ncap, d_lds, (u32)n_ld,
                     d_points, d_state_out);
  HIP_CHECK(hipGetLastError());
}

}  // namespace msamd
