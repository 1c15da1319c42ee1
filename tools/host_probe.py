"""Diagnostics: host-side probes (MSAMD_TRACE_HOST=1) of one HBM-resident and one host-resident proof, printed by the library."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
fe = pkg.frontend
ctx = pkg.Context(0)
system = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
traces, claims = fe.u32_add_bench_witness(1 << 20)
packed = fe.pack_claims(claims)
for name, w in (("HBM-resident", system.witness(traces, packed)), ("host-resident", system.host_witness(traces, packed))):
    for _ in range(6):
        system.prove_multiple_claims(w)
    ts = []
    for _ in range(10):
        t = time.perf_counter()
        system.prove_multiple_claims(w)
        ts.append(1e3 * (time.perf_counter() - t))
    print("==== %s: mean %.3f min %.3f ms" % (name, sum(ts) / len(ts), min(ts)), file=sys.stderr, flush=True)
    os.environ["MSAMD_TRACE_HOST"] = "1"
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
        system.prove_multiple_claims(w)
    del os.environ["MSAMD_TRACE_HOST"]
    del w
