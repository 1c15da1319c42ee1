"""Diagnostics: per-phase wall time of one proof of the bench workload (MSAMD_TRACE=1 adds syncs + prints)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

log_adds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pkg = load_package()
fe = pkg.frontend
ctx = pkg.Context(0)
system = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
traces, claims = fe.u32_add_bench_witness(1 << log_adds)
packed = fe.pack_claims(claims)
w = system.witness(traces, packed)
for i in range(3):
    system.prove_multiple_claims(w)
if len(sys.argv) > 2 and sys.argv[2] == "notrace":
    ctx.sync() if hasattr(ctx, "sync") else None
    sys.exit(0)
os.environ["MSAMD_TRACE"] = "1"
t = time.time()
p = system.prove_multiple_claims(w, want_times=True)
print("traced proof: %.3f ms" % (1e3 * (time.time() - t)), p.stage_ms)
