"""Diagnostics: a few joint proofs (ms_prove_sharded) with a single rank on the library's own RCCL transport (no torch),
for rocprofv3 timelines:  rocprofv3 --kernel-trace --memory-copy-trace -- python3 tools/joint_trace.py"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package

pkg = load_package()
fe = pkg.frontend
sharded = importlib.import_module("multi_stark_amd.sharded")
ctx = pkg.Context(0)
system = pkg.System.new(ctx, fe.bench_params(), fe.multi_u32_add_system_inputs(1))
traces, claims = fe.multi_u32_add_witness(1, 1 << 20)
packed = fe.pack_claims(claims)
w = system.witness(traces, packed)
comm = sharded.RcclComm(ctx, None, 0, 1)
owners = sharded.u32_add_owners(1)
for i in range(5):
    t = time.time()
    p = system.prove_sharded(w, comm, owners, want_times=False)
    print("joint proof %d: %.2f ms %s" % (i, 1e3 * (time.time() - t), p.stage_ms if i == 3 else ""), flush=True)
comm.close()
