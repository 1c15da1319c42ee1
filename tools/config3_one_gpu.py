"""BASELINE config 3 AS SPECIFIED, at full size, on ONE GPU: eight thread ranks (the library's in-process transport) prove
[ByteTable, U32Add x 8] at 2^log_adds additions per rank together; every rank's bytes must equal the single-GPU proof of the
same nine-circuit system, and the library's verifier must accept. The ranks time-share the card, so the wall time is about
eight proofs' worth of kernels plus the exchanges as HBM copies - a rehearsal of the code path and of the per-rank kernel
work at N = 8, not a scaling number.
  python tools/config3_one_gpu.py [log_adds] [world] [proofs]"""
import hashlib
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package

log_adds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
pkg = load_package()
fe = pkg.frontend
sharded = importlib.import_module("multi_stark_amd.sharded")
t0 = time.time()
traces, claims = fe.multi_u32_add_witness(world, 1 << log_adds)
packed = fe.pack_claims(claims)
owners = sharded.u32_add_owners(world)
print("witness: %d adders x 2^%d rows, %d claims, %.1f s" % (world, log_adds, len(claims), time.time() - t0), flush=True)
ctx0 = pkg.Context(0)
sys0 = pkg.System.new(ctx0, fe.bench_params(), fe.multi_u32_add_system_inputs(world))
full = sys0.witness(traces, packed)
want = sys0.prove_multiple_claims(full).to_bytes()
ctx0.sync()
t = time.perf_counter()
for _ in range(reps):
    sys0.prove_multiple_claims(full)
ctx0.sync()
single_ms = 1e3 * (time.perf_counter() - t) / reps
assert sys0.verify_multiple_claims(packed, want) == 0
print("single GPU, one proof of the %d-circuit system: %.2f ms, %d bytes, sha256 %s" % (world + 1, single_ms, len(want), hashlib.sha256(want).hexdigest()[:16]), flush=True)
del full
ctx0.trim()


def rank_body(rank, group):
    ctx = pkg.Context(0)
    system = pkg.System.new(ctx, fe.bench_params(), fe.multi_u32_add_system_inputs(world))
    mine = [t.copy() if owners[i] in (-1, rank) else None for i, t in enumerate(traces)]
    remote = {i: traces[i].shape[0] for i in range(len(traces)) if owners[i] not in (-1, rank)}
    w = system.host_witness(mine, packed, remote_heights=remote)
    comm = group.comm(ctx, rank)
    try:
        got = system.prove_sharded(w, comm, owners).to_bytes()
        assert got == want, "rank %d: joint proof differs from the single-GPU proof" % rank
        ctx.sync()
        t = time.perf_counter()
        for _ in range(reps):
            system.prove_sharded(w, comm, owners)
        ctx.sync()
        ms = 1e3 * (time.perf_counter() - t) / reps
        st = system.prove_sharded(w, comm, owners, want_times=True).stage_ms
        return ms, comm.bytes_moved, st
    finally:
        comm.close()


group = sharded.LocalGroup(world)
try:
    res = group.run(rank_body)
finally:
    group.close()
print("joint proof by %d thread ranks sharing the GPU: bytes identical to the single-GPU proof on every rank" % world)
print("  wall per joint proof %.2f ms (max over ranks; the ranks time-share one GPU: compare with %d x the per-rank work, single-GPU proof %.2f ms)" % (
    max(r[0] for r in res), world, single_ms))
print("  bytes through the exchanges per rank and proof: %.1f MB" % (res[0][1] / (reps + 2) / 1e6))
print("  stage_ms of rank 0 (wall, with %d ranks sharing the card): %s" % (world, {k: round(v, 2) for k, v in res[0][2].items()}))
