"""The joint prover's own overhead: ONE rank (world 1) proves [ByteTable, U32Add] at 2^20 additions through ms_prove_sharded -
on the RCCL transport and on the in-process transport - next to the plain prover (ms_prove) on the same witness, same
context, same box. With one rank no byte crosses a link, so the difference is what the sharded code path itself costs
(column groups, sub-tree + top levels, gathers, extra synchronisations) before any exchange is added.
  python tools/joint_vs_plain.py [log_adds] [proofs]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package

log_adds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pkg = load_package()
fe = pkg.frontend
sharded = importlib.import_module("multi_stark_amd.sharded")
ctx = pkg.Context(0)
system = pkg.System.new(ctx, fe.bench_params(), fe.multi_u32_add_system_inputs(1))
traces, claims = fe.multi_u32_add_witness(1, 1 << log_adds)
packed = fe.pack_claims(claims)
owners = sharded.u32_add_owners(1)
rccl = sharded.RcclComm(ctx, None, 0, 1)
group = sharded.LocalGroup(1)
local = group.comm(ctx, 0)


def timed(fn):
    for _ in range(3):
        fn()
    ctx.sync()
    best, tot = 1e9, 0.0
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        dt = 1e3 * (time.perf_counter() - t)
        best, tot = min(best, dt), tot + dt
    return tot / reps, best


out = {}
for name, w in (("hbm-resident", system.witness(traces, packed)), ("host-resident", system.host_witness(traces, packed))):
    want = system.prove_multiple_claims(w).to_bytes()
    assert system.prove_sharded(w, rccl, owners).to_bytes() == want and system.prove_sharded(w, local, owners).to_bytes() == want
    plain = timed(lambda: system.prove_multiple_claims(w))
    j_rccl = timed(lambda: system.prove_sharded(w, rccl, owners))
    j_local = timed(lambda: system.prove_sharded(w, local, owners))
    n0 = ctx.sync_count()
    system.prove_multiple_claims(w)
    n1 = ctx.sync_count()
    system.prove_sharded(w, local, owners)
    n2 = ctx.sync_count()
    print("   host synchronisations per proof: plain %d, joint %d" % (n1 - n0, n2 - n1))
    st = system.prove_sharded(w, rccl, owners, want_times=True).stage_ms
    sp = system.prove_multiple_claims(w, want_times=True).stage_ms
    print("%-14s plain %.3f ms (best %.3f) | joint/rccl %.3f (best %.3f, %+.3f ms, %+.1f %%) | joint/local %.3f (best %.3f, %+.3f ms)" % (
        name, plain[0], plain[1], j_rccl[0], j_rccl[1], j_rccl[0] - plain[0], 100 * (j_rccl[0] / plain[0] - 1), j_local[0], j_local[1],
        j_local[0] - plain[0]), flush=True)
    print("   stage_ms plain %s" % {k: round(v, 3) for k, v in sp.items()})
    print("   stage_ms joint %s" % {k: round(v, 3) for k, v in st.items()}, flush=True)
    del w
rccl.close()
local.close()
group.close()
