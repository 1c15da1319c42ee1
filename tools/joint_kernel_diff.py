"""Per-kernel-class device time (HIP events) of one plain proof and one joint proof at world 1 on the same witness: tells
whether the joint prover's overhead is kernel work or idle time.  python tools/joint_kernel_diff.py [log_adds]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package

log_adds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
pkg = load_package()
fe = pkg.frontend
sharded = importlib.import_module("multi_stark_amd.sharded")
ctx = pkg.Context(0)
system = pkg.System.new(ctx, fe.bench_params(), fe.multi_u32_add_system_inputs(1))
traces, claims = fe.multi_u32_add_witness(1, 1 << log_adds)
packed = fe.pack_claims(claims)
owners = sharded.u32_add_owners(1)
comm = sharded.RcclComm(ctx, None, 0, 1)
w = system.witness(traces, packed)
runs = {"plain": lambda: system.prove_multiple_claims(w), "joint": lambda: system.prove_sharded(w, comm, owners)}
tables = {}
for name, fn in runs.items():
    for _ in range(3):
        fn()
    ctx.set_profile(ctx.kernel_names())
    ctx.reset_stats()
    k = 5
    for _ in range(k):
        fn()
    tables[name] = {n: (s["launches"] / k, s["ms"] / k) for n, s in ctx.kernel_stats().items()}
    ctx.set_profile([])
    ctx.sync()
    t = time.perf_counter()
    for _ in range(10):
        fn()
    ctx.sync()
    tables[name]["_wall"] = (0, 1e3 * (time.perf_counter() - t) / 10)
print("%-16s %8s %8s %8s   (launches plain / joint)" % ("class", "plain", "joint", "diff"))
tp = tj = 0.0
for n in ctx.kernel_names():
    (lp, mp), (lj, mj) = tables["plain"][n], tables["joint"][n]
    if lp or lj:
        print("%-16s %8.3f %8.3f %+8.3f   %g / %g" % (n, mp, mj, mj - mp, lp, lj))
        tp, tj = tp + mp, tj + mj
print("%-16s %8.3f %8.3f %+8.3f" % ("sum of kernels", tp, tj, tj - tp))
print("%-16s %8.3f %8.3f %+8.3f" % ("wall (no events)", tables["plain"]["_wall"][1], tables["joint"]["_wall"][1], tables["joint"]["_wall"][1] - tables["plain"]["_wall"][1]))
comm.close()
