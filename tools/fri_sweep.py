"""Diagnostics: HBM-resident proof time for several values of an environment switch.  python tools/fri_sweep.py VAR v1 v2 ..."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

var, vals = sys.argv[1], sys.argv[2:]
pkg = load_package()
fe = pkg.frontend
ctx = pkg.Context(0)
system = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
traces, claims = fe.u32_add_bench_witness(1 << 20)
w = system.witness(traces, fe.pack_claims(claims))
ref = None
for rep in range(2):
    for v in vals:
        if v == "-":
            os.environ.pop(var, None)
        else:
            os.environ[var] = v
        for _ in range(4):
            p = system.prove_multiple_claims(w).to_bytes()
        ref = ref or p
        assert p == ref, "proof bytes changed with %s=%s" % (var, v)
        ctx.sync()
        t = time.perf_counter()
        for _ in range(20):
            system.prove_multiple_claims(w)
        ctx.sync()
        ms = 1e3 * (time.perf_counter() - t) / 20
        st = system.prove_multiple_claims(w, want_times=True).stage_ms
        print("%s=%-4s %.3f ms per proof   fri_open %.3f" % (var, v, ms, st["fri_open"]), flush=True)
