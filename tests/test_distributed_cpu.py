"""CPU tests of the N > 1 path (gloo): (1) world 2, replicas mode: per-rank witness seeds, independent proofs (the oracle
stands in for the device here - this exercises the sharding / gather logic, not the kernels), all_gather of the commitments
and the joint digest on every rank; (2) world 2 and 4, the joint prover's exchange patterns: every callback of the ms_comm
table (multi-stark_amd/sharded.py::TorchComm) on HOST buffers - row-range all-to-all (packed, strided start / wait, straight
out of a column-major matrix), the per-matrix hand-out of the general ownership pattern, all_gather - against what the
pattern is specified to deliver in include/mstark.h."""
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


def _exchange_worker(rank, world, port, q):
    """every callback of the ms_comm table over gloo on host buffers; value at (owner o, column c, row r) = o * 2^40 + c * 2^20 + r"""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        sys.path.insert(0, ROOT)
        import ctypes as C
        import importlib

        from __graft_entry__ import load_package

        load_package()
        sharded = importlib.import_module("multi_stark_amd.sharded")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            comm = sharded.TorchComm(memory="host")
            t = comm.struct
            N, h, w = world, 16 * world, 5
            rows = h // N

            def ptr(a):
                return C.c_void_p(a.ctypes.data)

            def val(o, c, r):
                return (o << 40) + (c << 20) + r

            # this rank's "LDE": column-major h x w, element (r, c) at c * h + r
            mine = np.array([[val(rank, c, r) for r in range(h)] for c in range(w)], dtype=np.uint64)
            # --- all_to_all_cols_start: rows [k h / N, (k + 1) h / N) of every column go to rank k
            recv = np.zeros((N, w, rows), dtype=np.uint64)
            assert t.all_to_all_cols_start(None, ptr(mine), rows * 8, h * 8, ptr(recv), w * rows * 8, rows * 8, w, rows * 8) == 0
            assert t.all_to_all_wait(None) == 0
            want = np.array([[[val(o, c, rank * rows + r) for r in range(rows)] for c in range(w)] for o in range(N)], dtype=np.uint64)
            assert np.array_equal(recv, want), "all_to_all_cols_start"
            # --- the same in two column groups (what the prover does while it transforms the next group)
            recv2 = np.zeros((N, w, rows), dtype=np.uint64)
            for c0, c1 in ((0, 2), (2, w)):
                assert t.all_to_all_cols_start(None, C.c_void_p(mine.ctypes.data + c0 * h * 8), rows * 8, h * 8,
                                               C.c_void_p(recv2.ctypes.data + c0 * rows * 8), w * rows * 8, rows * 8, c1 - c0, rows * 8) == 0
            assert t.all_to_all_wait(None) == 0 and np.array_equal(recv2, want), "column groups"
            # --- packed all_to_all and the strided start / wait pair
            send = np.ascontiguousarray(np.stack([mine[:, k * rows:(k + 1) * rows] for k in range(N)]))  # [peer][column][row]
            recv3 = np.zeros_like(send)
            assert t.all_to_all(None, ptr(send), ptr(recv3), w * rows * 8) == 0 and np.array_equal(recv3, want), "all_to_all"
            recv4 = np.zeros_like(send)
            for c0, c1 in ((0, 3), (3, w)):
                assert t.all_to_all_start(None, C.c_void_p(send.ctypes.data + c0 * rows * 8), w * rows * 8,
                                          C.c_void_p(recv4.ctypes.data + c0 * rows * 8), w * rows * 8, (c1 - c0) * rows * 8) == 0
            assert t.all_to_all_wait(None) == 0 and np.array_equal(recv4, want), "all_to_all_start"
            # --- scatter_cols_start: one rank's matrix handed out by row ranges (general ownership), every root in turn
            for root in range(N):
                got = np.full((w, rows), 7, dtype=np.uint64)
                assert t.scatter_cols_start(None, root, ptr(mine) if rank == root else None, rows * 8, h * 8,
                                            None if rank == root else ptr(got), rows * 8, w, rows * 8) == 0
                assert t.all_to_all_wait(None) == 0
                if rank != root:
                    assert np.array_equal(got, want[root]), "scatter from %d" % root
                else:
                    assert (got == 7).all()  # nothing is written on the root: its rows stay where they are
            # --- all_gather
            part = np.array([val(rank, 1, 2), val(rank, 3, 4)], dtype=np.uint64)
            allp = np.zeros((N, 2), dtype=np.uint64)
            assert t.all_gather(None, ptr(part), ptr(allp), 16) == 0
            assert np.array_equal(allp, np.array([[val(o, 1, 2), val(o, 3, 4)] for o in range(N)], dtype=np.uint64))
            comm.reraise()
            q.put((rank, "ok", comm.bytes_moved))
        finally:
            dist.barrier()
            dist.destroy_process_group()
    except BaseException as e:
        q.put((rank, "ERROR: %r" % (e,), 0))
        raise


@pytest.mark.parametrize("world", [2, 4])
def test_exchange_patterns_of_the_joint_prover(world):
    port = 29300 + (os.getpid() % 2000) + world
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
    assert all(p.exitcode == 0 for p in procs)
    assert all(r[2] > 0 for r in res)
