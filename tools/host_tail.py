"""Diagnostics: host time per proof outside the GPU's span (MSAMD_TRACE_HOST=1 prints the library's own probes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package(); fe = pkg.frontend
ctx = pkg.Context(0)
system = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
traces, claims = fe.u32_add_bench_witness(1 << 20)
w = system.witness(traces, fe.pack_claims(claims))
for i in range(5):
    system.prove_multiple_claims(w)
os.environ["MSAMD_TRACE_HOST"] = "1"
for i in range(3):
    t = time.perf_counter()
    system.prove_multiple_claims(w)
    print("python call: %.1f us" % (1e6 * (time.perf_counter() - t)), file=sys.stderr)
