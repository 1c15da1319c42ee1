"""Large-trace stress run of the bench workload (BASELINE config 5 = 2^26 additions): prove, time, verify.

usage: python3 tools/stress.py LOG_ADDS [--no-verify] [--proofs N] [--airs K]
--airs K proves the system [ByteTable, U32Add x K] (config 3 as ONE proof, here on one GPU), 2^LOG_ADDS additions per AIR.
The oracle is used only as the checker (its verifier accepts or rejects the proof bytes)."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402


def mem_gb():
    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    return (total.value - free.value) / 2**30, total.value / 2**30


def say(*a):
    print("[stress %7.1fs]" % (time.time() - T0), *a, flush=True)


T0 = time.time()
log_adds = int(sys.argv[1])
verify = "--no-verify" not in sys.argv
n_proofs = int(sys.argv[sys.argv.index("--proofs") + 1]) if "--proofs" in sys.argv else 2
pkg = load_package()
fe = pkg.frontend
ctx = pkg.Context(0)
airs = int(sys.argv[sys.argv.index("--airs") + 1]) if "--airs" in sys.argv else 0
inputs = fe.multi_u32_add_system_inputs(airs) if airs else fe.u32_add_system_inputs()
system = pkg.System.new(ctx, fe.bench_params(), inputs)
say("system ready (%d circuits); generating witness for 2^%d additions per AIR" % (len(inputs), log_adds))
if airs:
    traces, claims = fe.multi_u32_add_witness(airs, 1 << log_adds)
else:
    traces, claims = fe.u32_add_bench_witness(1 << log_adds)
packed = fe.pack_claims(claims)
say("witness on host: traces %.2f GB, claims %.2f GB" % (sum(t.nbytes for t in traces) / 1e9, packed[1].nbytes / 1e9))
w = system.witness(traces, packed)
ctx.sync()
say("witness resident in HBM; device memory in use %.1f of %.1f GiB" % mem_gb())
proof = None
for i in range(n_proofs):
    t = time.time()
    p = system.prove_multiple_claims(w, want_times=True)
    dt = time.time() - t
    rows = max(airs, 1) * (1 << log_adds) + 256
    say("proof %d: %.1f ms wall (%.1f M rows/s), %d bytes, stages %s; device memory %.1f GiB" % (
        i, 1e3 * dt, rows / dt / 1e6, len(p.to_bytes()), {k: round(v, 1) for k, v in p.stage_ms.items()}, mem_gb()[0]))
    if proof is not None and p.to_bytes() != proof:
        say("FAIL: proofs differ between runs")
        sys.exit(1)
    proof = p.to_bytes()
if verify:
    import oracle as orc  # tests/oracle.py: ctypes wrapper of the CPU restatement
    o = orc.System(system.blob)
    t = time.time()
    rc = o.verify(packed, proof)
    say("oracle verifier: rc=%d in %.1f s" % (rc, time.time() - t))
    if rc != 0:
        sys.exit(1)
    bad = claims.copy()
    bad[len(bad) // 3, 3] ^= 1
    rc = o.verify(fe.pack_claims(bad), proof)
    say("oracle verifier on a flipped claim: rc=%d (must be non-zero)" % rc)
    if rc == 0:
        sys.exit(1)
say("OK")
