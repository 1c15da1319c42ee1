// Compiled-circuit node program as it crosses into the quotient kernel (mirrors graph::Node,
// /root/reference/src/graph.rs:35-46). Op codes equal the node kinds of the system blob.
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

#include "msamd.h"

namespace msamd {

enum : uint32_t { OP_CONST = 0, OP_VAR, OP_PUBLIC, OP_IS_FIRST, OP_IS_LAST, OP_IS_TRANS, OP_ADD, OP_SUB, OP_MUL, OP_NEG };

struct PNode {
  uint32_t kind = 0, source = 0, offset = 0;
  uint64_t a = 0, b = 0;
};

void build_program(Ctx& ctx, const std::vector<PNode>& nodes, const std::vector<uint32_t>& zeros,
                   const std::vector<std::pair<uint32_t, std::vector<uint32_t>>>& lookups, DProgram& out);

// quotient_jit.hip: compile (or fetch from the cache) the circuit's own quotient kernel; launch it
struct QParams;
void quotient_jit_build(const std::vector<PNode>& nodes, const std::vector<uint32_t>& zeros,
                        const std::vector<std::pair<uint32_t, std::vector<uint32_t>>>& lookups, size_t quotient_degree, JitKernel& out);
void quotient_jit_launch(Ctx& ctx, const JitKernel& k, const QParams& p, size_t nq);
// the same for the BabyBear / Ext4 configuration (argument block: msbb::QuotArgs of bb_quotient_params.h, passed as bytes)
void bb_quotient_jit_build(const std::vector<PNode>& nodes, const std::vector<uint32_t>& zeros,
                           const std::vector<std::pair<uint32_t, std::vector<uint32_t>>>& lookups, JitKernel& out);
void bb_quotient_jit_launch(Ctx& ctx, const JitKernel& k, const void* args, size_t args_size, size_t rows);
// the circuit's stage-2 terms kernel (messages, batch inverse, mult / message) for its list of argument counts
struct Stage2Params;
struct Stage2TraceParams;
void stage2_jit_build(const std::vector<uint32_t>& arg_counts, JitKernel& out);
void stage2_jit_launch(Ctx& ctx, const JitKernel& k, const Stage2Params& p);
// the terms pass fed by the trace (lookup expressions evaluated in the kernel); empty `out` = not available for this circuit
void stage2_trace_jit_build(const std::vector<PNode>& nodes, const std::vector<std::pair<uint32_t, std::vector<uint32_t>>>& lookups,
                            size_t main_w, size_t pre_w, size_t prefix_len, JitKernel& out);
void stage2_trace_jit_launch(Ctx& ctx, const JitKernel& k, const Stage2TraceParams& p);

// lookup values of SystemWitness::from_stage_1 on the device; false = prefix too large for the LDS slot file
bool lookup_values_device(Ctx& ctx, const DProgram& prefix, const u64* d_trace, const u64* d_pre, size_t h, size_t main_w,
                          size_t pre_w, size_t args_w, u64* d_mult, u64* d_args, hipStream_t on_stream = nullptr);

}  // namespace msamd
