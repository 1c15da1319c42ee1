"""Throughput of a prover SERVICE: K proofs in flight at once on one GPU (one host thread and one ms_ctx each), against
one proof at a time. While one proof sits in a host round trip or in FRI's latency-bound rounds, the other one's
transforms fill the chip.   python tools/two_in_flight.py [log_adds] [proofs per thread] [threads ...]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

log_adds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
counts = [int(x) for x in sys.argv[3:]] or [1, 2, 3]
host = os.environ.get("TIF_HOST") == "1"
pkg = load_package()
fe = pkg.frontend
traces, claims = fe.u32_add_bench_witness(1 << log_adds)
packed = fe.pack_claims(claims)
slots = []
for k in range(max(counts)):
    ctx = pkg.Context(0)
    system = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    w = system.host_witness([t.copy() for t in traces], packed) if host else system.witness(traces, packed)
    ref = system.prove_multiple_claims(w).to_bytes()
    slots.append((ctx, system, w, ref))
assert all(s[3] == slots[0][3] for s in slots)
for k in counts:
    def body(i):
        ctx, system, w, ref = slots[i]
        for _ in range(3):
            system.prove_multiple_claims(w)
        barrier.wait()
        for _ in range(n):
            p = system.prove_multiple_claims(w)
        assert p.to_bytes() == ref
        ctx.sync()
    barrier = threading.Barrier(k + 1)
    ths = [threading.Thread(target=body, args=(i,)) for i in range(k)]
    for t in ths:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    print("%d in flight (%s witness): %.3f ms per proof (%.1f M rows/s), %.3f ms latency per proof" % (
        k, "host-resident" if host else "HBM-resident", 1e3 * dt / (k * n), (256 + (1 << log_adds)) * k * n / dt / 1e6, 1e3 * dt / n), flush=True)
