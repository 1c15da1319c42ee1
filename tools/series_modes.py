import os, sys, time
sys.path.insert(0, '/root/repo')
from __graft_entry__ import load_package
pkg = load_package(); fe = pkg.frontend
ctx = pkg.Context(0)
system = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
traces, claims = fe.u32_add_bench_witness(1 << 20)
packed = fe.pack_claims(claims)
mode = sys.argv[1]
w = system.witness(traces, packed) if mode == "hbm" else system.host_witness(traces, packed)
for _ in range(5): system.prove_multiple_claims(w)
ts = []
for _ in range(200):
    t = time.perf_counter(); system.prove_multiple_claims(w); ts.append(1e3 * (time.perf_counter() - t))
ts2 = sorted(ts)
print(mode, "mean %.3f median %.3f p90 %.3f max %.3f  spikes>8.5ms: %d" % (sum(ts)/len(ts), ts2[100], ts2[180], ts2[-1], sum(1 for x in ts if x > 8.5)))
