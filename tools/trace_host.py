"""Diagnostics: a few proofs from a HOST-resident witness (upload inside the proof), for rocprofv3 --kernel-trace --memory-copy-trace."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

log_adds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pkg = load_package()
fe = pkg.frontend
ctx = pkg.Context(0)
system = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
traces, claims = fe.u32_add_bench_witness(1 << log_adds)
w = system.host_witness(traces, fe.pack_claims(claims))
for i in range(4):
    system.prove_multiple_claims(w)
ctx.sync()
