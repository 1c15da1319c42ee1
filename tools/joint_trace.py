"""Diagnostics: a few joint proofs (ms_prove_sharded) with a single rank over RCCL, for rocprofv3 timelines.
Run as: RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29531 python3 tools/joint_trace.py"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from __graft_entry__ import load_package

pkg = load_package()
fe = pkg.frontend
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
sharded = importlib.import_module("multi_stark_amd.sharded")
ctx = pkg.Context(0)
system = pkg.System.new(ctx, fe.bench_params(), fe.multi_u32_add_system_inputs(1))
traces, claims = fe.multi_u32_add_witness(1, 1 << 20)
packed = fe.pack_claims(claims)
w = system.witness(traces, packed)
comm = sharded.TorchComm(0)
owners = sharded.u32_add_owners(1)
for i in range(4):
    t = time.time()
    p = system.prove_sharded(w, comm, owners)
    print("joint proof %d: %.2f ms" % (i, 1e3 * (time.time() - t)), flush=True)
dist.destroy_process_group()
