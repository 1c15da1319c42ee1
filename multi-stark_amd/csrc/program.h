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

}  // namespace msamd
