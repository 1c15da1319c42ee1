"""Diagnostics: wall time of each of N consecutive host-resident proofs in one process (how steady is the narrow upload?)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
if len(sys.argv) > 2 and sys.argv[2] == "torch":
    import torch  # noqa: F401  (does the import alone change anything?)
pkg = load_package()
fe = pkg.frontend
ctx = pkg.Context(0)
system = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
traces, claims = fe.u32_add_bench_witness(1 << 20)
w = system.host_witness(traces, fe.pack_claims(claims))
for _ in range(5):
    system.prove_multiple_claims(w)
ts = []
for _ in range(n):
    t = time.perf_counter()
    system.prove_multiple_claims(w)
    ts.append(1e3 * (time.perf_counter() - t))
print("mean %.3f  min %.3f  max %.3f" % (sum(ts) / n, min(ts), max(ts)))
print(" ".join("%.2f" % x for x in ts))
