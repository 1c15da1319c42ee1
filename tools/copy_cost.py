import os, sys, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
import numpy as np
pkg = load_package(); fe = pkg.frontend
ctx = pkg.Context(0)
system = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
traces, claims = fe.u32_add_bench_witness(1 << 20)
w = system.witness(traces, fe.pack_claims(claims))
for _ in range(5): p = system.prove_multiple_claims(w)
n = len(p.to_bytes())
out, cap = system._out_buffer()
ts = []
for _ in range(30):
    t0 = time.perf_counter(); s = C.string_at(out.ctypes.data, n); t1 = time.perf_counter(); ts.append(t1 - t0); del s
print("proof bytes", n, "string_at: median %.1f us, min %.1f" % (1e6 * sorted(ts)[15], 1e6 * min(ts)))
a = np.empty(n, dtype=np.uint8); b = np.empty(n, dtype=np.uint8); ts = []
for _ in range(30):
    t0 = time.perf_counter(); np.copyto(b, a); ts.append(time.perf_counter() - t0)
print("warm memcpy of the same size: median %.1f us" % (1e6 * sorted(ts)[15]))
ts = []
for _ in range(30):
    t0 = time.perf_counter(); system.prove_multiple_claims(w); ts.append(time.perf_counter() - t0)
print("prove_multiple_claims: median %.1f us" % (1e6 * sorted(ts)[15]))
