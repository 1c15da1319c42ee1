"""CPU test of the N > 1 path (gloo, world_size 2): per-rank witness seeds, independent proofs (oracle stands in for
the device here — this test exercises the sharding/gather logic, not the kernels), all_gather of the commitments and
the joint digest on every rank."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2", OMP_WAIT_POLICY="passive")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from __graft_entry__ import load_package
    import importlib
    import oracle

    pkg = load_package()
    fe = pkg.frontend
    mgpu = importlib.import_module("multi_stark_amd.distributed")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        a0, b0 = mgpu.rank_seeds(rank)
        traces, claims = fe.u32_add_bench_witness(1 << 6, a0, b0)
        comp = [fe.compile_circuit(ci) for ci in fe.u32_add_system_inputs()]
        s = oracle.System(fe.system_blob(fe.test_params(), comp))
        packed = fe.pack_claims(claims)
        proof = s.prove(traces, packed)
        assert s.verify(packed, proof) == 0
        mine = mgpu.commitments_of(proof, 2)
        allc = mgpu.gather_commitments(mine)
        assert len(allc) == world and allc[rank] == mine and all(len(c) == 96 for c in allc)
        # the off-thread gatherer bench.py uses: submissions are gathered in order, finish() waits for all of them
        g = mgpu.CommitmentGatherer(len(mine))
        second = bytes(reversed(mine))
        g.submit(mine)
        g.submit(second)
        got = g.finish()
        assert got[0] == allc and got[1][rank] == second and g.finish() == []
        g.close()
        q.put((rank, mine.hex(), mgpu.joint_digest(allc).hex()))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_two_rank_commitment_gather(oracle):
    world, port = 2, 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # distinct per-rank witnesses -> distinct commitments; every rank derives the same joint digest
    assert res[0][1] != res[1][1]
    assert res[0][2] == res[1][2]
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    import importlib

    load_package()
    mgpu = importlib.import_module("multi_stark_amd.distributed")
    assert mgpu.joint_digest([bytes.fromhex(res[0][1]), bytes.fromhex(res[1][1])]).hex() == res[0][2]
    assert mgpu.rank_seeds(0) == (0xDEADBEEF, 0xCAFEBABE)
