// Argument types of the stage-2 kernels, shared by lookup.hip and by the per-circuit stage-2 kernel that
// quotient_jit.hip generates (this header is on hiprtc's include path).
#pragma once
#include "gl_dev.h"

namespace msamd {

constexpr int MAX_GPOW = 64;    // gamma powers kept in a kernel argument; longer argument lists fall back to Horner

struct GammaPows {
  E2 g[MAX_GPOW];
  u32 n;  // number of valid powers (gamma^0 .. gamma^(n-1))
};

// beta, gamma and gamma's powers live in DEVICE memory, not in the argument blocks: the host writes them (h2d) when it knows
// the challenges, or the device transcript does (outer.hip) - stage 2 is then queued before beta has reached the host
struct ChallengeBG {
  E2 beta, gamma;
  GammaPows gp;
};
GL_HD void challenge_bg_fill(ChallengeBG& c, E2 beta, E2 gamma) {
  c.beta = beta;
  c.gamma = gamma;
  c.gp.n = MAX_GPOW;
  E2 g = e2(1);
  for (int i = 0; i < MAX_GPOW; i++) {
    c.gp.g[i] = g;
    g = e2_mul(g, gamma);
  }
}

struct Stage2Params {
  const u64* mult;   // n x L row-major
  const u64* args;   // n x args_width row-major
  size_t n;
  const ChallengeBG* ch;
  E2* terms;         // n x L: mult / message
  E2* rowsum;        // n
};

// the same pass fed by the TRACE: the circuit's lookup expressions (the prefix of its node program) are evaluated in the
// kernel, so that no LookupValues are materialised (host-resident witnesses: src/system.rs:244-328 fused into
// src/lookup.rs:472-543)
struct Stage2TraceParams {
  const u64* trace;  // n x main_w row-major, as uploaded
  const u64* pre;    // n x pre_w row-major (or null)
  size_t n;
  const ChallengeBG* ch;
  E2* terms;
  E2* rowsum;
};

}  // namespace msamd
