"""GPU parity of the multi-rank prover (ms_prove_sharded): `world` ranks prove the system [ByteTable, U32Add x world]
together; every rank must return exactly the bytes the single-GPU prover produces for the same system and witness, and the
oracle verifier must accept them. The ranks share the one GPU of the test box, in two forms: processes over gloo
(`TorchComm`, world <= 4: the box allows six processes on its card) and threads of the test process over the library's
in-process transport (`LocalGroup`, csrc/comm_local.hip: world 8 = BASELINE config 3 as specified, and the same
stream-ordered, non-blocking column exchange the RCCL transport offers, here with real peers)."""
import hashlib
import os
import sys

import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, log_adds, variant, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="4", OMP_WAIT_POLICY="passive")
        if variant == "pack":  # the exchange through a packed send buffer (what a transport without all_to_all_cols_start gets)
            os.environ["MSAMD_SHARDED_PACK"] = "1"
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import importlib

        import torch.distributed as dist
        from __graft_entry__ import load_package

        pkg = load_package()
        fe = pkg.frontend
        sharded = importlib.import_module("multi_stark_amd.sharded")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            ctx = pkg.Context(0)
            params = fe.bench_params()
            if variant == "cap1":  # commitments are 2-digest caps, 4-coefficient final polynomial, host-driven FRI rounds
                params = fe.Params(log_blowup=2, cap_height=1, log_final_poly_len=2, num_queries=20, commit_proof_of_work_bits=3,
                                   query_proof_of_work_bits=5)
            system = pkg.System.new(ctx, params, fe.multi_u32_add_system_inputs(world))
            traces, claims = fe.multi_u32_add_witness(world, 1 << log_adds)
            packed = fe.pack_claims(claims)
            owners = sharded.u32_add_owners(world)
            mine = [t if owners[i] in (-1, rank) else None for i, t in enumerate(traces)]
            remote = {i: traces[i].shape[0] for i in range(len(traces)) if owners[i] not in (-1, rank)}
            w = system.witness(mine, packed, remote_heights=remote)
            comm = sharded.TorchComm(0)
            proof = system.prove_sharded(w, comm, owners).to_bytes()
            # the same from a HOST-resident witness: this rank's traces and its slice of the claims are uploaded inside the proof
            hw = system.host_witness(mine, packed, remote_heights=remote)
            for _ in range(2):
                assert system.prove_sharded(hw, comm, owners).to_bytes() == proof, "host-resident sharded proof differs"
            del hw
            again = system.prove_sharded(w, comm, owners, want_times=True)
            assert again.to_bytes() == proof and again.stage_ms["total"] > 0
            if remote:
                with pytest.raises(pkg.MstarkError):
                    system.prove_multiple_claims(w)  # a witness with remote circuits is refused by the one-GPU prover
            if variant == "bad-owners" and world > 1:
                # every rank passes the same wrong map, so all of them fail before the first exchange
                with pytest.raises(pkg.MstarkError):
                    system.prove_sharded(w, comm, [-1] + list(reversed(range(world))))
            if rank == 0:
                import oracle

                full = system.witness(traces, packed)
                want = system.prove_multiple_claims(full).to_bytes()
                assert proof == want, "sharded proof differs from the single-GPU proof"
                assert oracle.System(system.blob).verify(packed, proof) == 0
                assert system.verify_multiple_claims(packed, proof) == 0
            q.put((rank, hashlib.sha256(proof).hexdigest(), comm.bytes_moved))
        finally:
            dist.barrier()
            dist.destroy_process_group()
    except BaseException as e:  # surface the failure in the parent instead of a queue timeout
        q.put((rank, "ERROR: %r" % (e,), 0))
        raise


def _random_worker(rank, world, port, seed, n_cases, q):
    """random systems: 0-2 replicated random circuits (any position) + one random circuit instantiated once per rank"""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="4", OMP_WAIT_POLICY="passive",
                          MSAMD_NO_JIT="1")
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import importlib

        import numpy as np
        import torch.distributed as dist
        from __graft_entry__ import load_package

        import fuzz_parity as fz

        pkg = load_package()
        fe = pkg.frontend
        sharded = importlib.import_module("multi_stark_amd.sharded")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            ctx = pkg.Context(0)
            comm = sharded.TorchComm(0)
            master = np.random.default_rng(seed)
            done = 0
            for case in range(n_cases):
                rng = np.random.default_rng(master.integers(0, 1 << 62))   # the same stream on every rank
                lb = int(rng.integers(1, 3))
                lw = world.bit_length() - 1
                params = fe.Params(log_blowup=lb, cap_height=int(rng.integers(0, lw + 1)), log_final_poly_len=0,
                                   num_queries=int(rng.integers(1, 12)), commit_proof_of_work_bits=int(rng.integers(0, 5)),
                                   query_proof_of_work_bits=int(rng.integers(0, 5)))
                shard_ci, shard_w, fixed_h = fz.random_circuit(rng, fe, lb)
                shard_h = fixed_h if fixed_h else 1 << int(rng.integers(2, 10))
                if shard_h < 4:
                    continue
                circuits, traces, owners = [], [], []
                n_rep = int(rng.integers(0, 3))
                rep_positions = sorted(int(x) for x in rng.integers(0, world + 1, n_rep))
                k = 0
                for slot in range(world + 1):
                    for _ in range(rep_positions.count(slot)):
                        ci, w, fh = fz.random_circuit(rng, fe, lb)
                        h = fh if fh else 1 << int(rng.integers(2, 10))
                        if h < 4:
                            h = 4 if not fh else h
                        circuits.append(ci)
                        traces.append(fz.rand_field(rng, (h, w)))
                        owners.append(-1)
                    if slot < world:
                        circuits.append(shard_ci)
                        traces.append(fz.rand_field(rng, (shard_h, shard_w)))
                        owners.append(k)
                        k += 1
                if any(t.shape[0] < 4 for t in traces):
                    continue
                claims = [[int(x) for x in fz.rand_field(rng, int(rng.integers(0, 5)))] for _ in range(int(rng.integers(0, 4)))]
                packed = fe.pack_claims(claims)
                try:
                    compiled = [fe.compile_circuit(c) for c in circuits]
                except fe.CompileError:
                    continue
                try:
                    system = pkg.System(ctx, fe.system_blob(params, compiled), len(compiled))
                except pkg.MstarkError:
                    continue   # e.g. a random constraint above the degree bound: every rank skips alike
                mine = [t if owners[i] in (-1, rank) else None for i, t in enumerate(traces)]
                remote = {i: traces[i].shape[0] for i in range(len(traces)) if owners[i] not in (-1, rank)}
                w_s = system.witness(mine, packed, remote_heights=remote)
                proof = system.prove_sharded(w_s, comm, owners).to_bytes()
                want = system.prove_multiple_claims(system.witness(traces, packed)).to_bytes()
                assert proof == want, "random case %d: sharded proof differs from the single-GPU proof" % case
                done += 1
            q.put((rank, "cases=%d" % done, comm.bytes_moved))
        finally:
            dist.barrier()
            dist.destroy_process_group()
    except BaseException as e:
        q.put((rank, "ERROR: %r" % (e,), 0))
        raise


@pytest.mark.parametrize("world,seed", [(2, 5), (4, 6)])
def test_sharded_random_systems(world, seed):
    port = 29700 + (os.getpid() % 1000) + world
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    n_cases = int(os.environ.get("MSAMD_SHARDED_FUZZ_CASES", "40"))  # more for an ad-hoc soak
    procs = [mpc.Process(target=_random_worker, args=(r, world, port, seed, n_cases, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    assert all(not str(r[1]).startswith("ERROR") for r in res), res
    assert all(p.exitcode == 0 for p in procs)
    assert len({r[1] for r in res}) == 1 and int(res[0][1].split("=")[1]) >= 15, res
    print("sharded random systems:", res[0][1], "world", world)


# 2^8 additions: the adders' LDE is as tall as the byte table's (same leaf group); 2^10: the byte table is injected
@pytest.mark.parametrize("world,log_adds,variant", [(1, 9, "bench"), (2, 8, "bench"), (2, 10, "bad-owners"), (4, 10, "bench"),
                                                    (2, 9, "cap1"), (2, 9, "pack"), (4, 8, "pack")])
def test_sharded_proof_equals_single_gpu_proof(world, log_adds, variant):
    port = 29600 + (os.getpid() % 1000) + 7 * world + log_adds + (3 if variant == "cap1" else 50 if variant == "pack" else 0)
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_worker, args=(r, world, port, log_adds, variant, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    assert all(not str(r[1]).startswith("ERROR") for r in res), res
    assert all(p.exitcode == 0 for p in procs)
    assert len({r[1] for r in res}) == 1  # every rank holds the same proof bytes
    if world > 1:
        assert all(r[2] > 0 for r in res)


# The library's own RCCL transport (ms_comm_rccl_*, csrc/comm_rccl.hip). One GPU on the test box allows world = 1 only
# (RCCL refuses two ranks on one device): the call path, handle lifetimes and the proof bytes are checked here, the
# multi-rank exchange pattern by the callback tests above, and 2 .. 8 real ranks by bench.py on the driver's node.
# Variants: the transport ordered with the prover's stream by events (the default: no host waits around the exchanges), by
# the host (MSAMD_SHARDED_HOST_SYNC), and the exchange through a packed send buffer (MSAMD_SHARDED_PACK); at 2^12 and at 2^16
# rows (column groups overlapping the transforms), from a device- and from a host-resident witness, several proofs in a row.
@pytest.mark.parametrize("var", ["", "MSAMD_SHARDED_HOST_SYNC", "MSAMD_SHARDED_PACK", "bypass"])
@pytest.mark.parametrize("log_adds", [12, 16])
def test_native_rccl_transport_world_1(pkg, ctx, oracle, fe, var, log_adds, monkeypatch):
    import importlib

    # a single rank normally never enters the transport (a gather from oneself is a copy on the prover's stream);
    # MSAMD_SHARDED_WORLD1_TRANSPORT=1 sends every exchange through it, which is what this test is about
    if var != "bypass":
        monkeypatch.setenv("MSAMD_SHARDED_WORLD1_TRANSPORT", "1")
    if var and var != "bypass":
        monkeypatch.setenv(var, "1")
    sharded = importlib.import_module("multi_stark_amd.sharded")
    system = pkg.System.new(ctx, fe.bench_params(), fe.multi_u32_add_system_inputs(1))
    traces, claims = fe.multi_u32_add_witness(1, 1 << log_adds)
    packed = fe.pack_claims(claims)
    w = system.witness(traces, packed)
    want = system.prove_multiple_claims(w).to_bytes()
    comm = sharded.RcclComm(ctx, None, 0, 1)
    got = system.prove_sharded(w, comm, sharded.u32_add_owners(1)).to_bytes()
    assert got == want
    hw = system.host_witness(traces, packed)
    for _ in range(3):
        assert system.prove_sharded(hw, comm, sharded.u32_add_owners(1)).to_bytes() == want
        assert system.prove_sharded(w, comm, sharded.u32_add_owners(1)).to_bytes() == want
    assert system.prove_multiple_claims(w).to_bytes() == want   # the plain prover on the same context afterwards
    assert (comm.bytes_moved > 0) == (var != "bypass")
    assert oracle.System(system.blob).verify(packed, got) == 0
    comm.close()
    with pytest.raises(pkg.MstarkError):
        sharded.RcclComm(ctx, None, 3, 2)  # rank out of range


_RCCL_SELF_SCRIPT = r"""
import importlib, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from __graft_entry__ import load_package
pkg = load_package(); fe = pkg.frontend
sharded = importlib.import_module("multi_stark_amd.sharded")
ctx = pkg.Context(0)
for params, log_adds in ((fe.bench_params(), 12), (fe.Params(log_blowup=2, cap_height=0, log_final_poly_len=2, num_queries=20, commit_proof_of_work_bits=3,
                                                             query_proof_of_work_bits=5), 10), (fe.bench_params(), 16)):
    system = pkg.System.new(ctx, params, fe.multi_u32_add_system_inputs(1))
    traces, claims = fe.multi_u32_add_witness(1, 1 << log_adds)
    packed = fe.pack_claims(claims)
    w = system.witness(traces, packed)
    want = system.prove_multiple_claims(w).to_bytes()
    comm = sharded.RcclComm(ctx, sharded.RcclComm.unique_id(), 0, 1)
    for _ in range(3):
        assert system.prove_sharded(w, comm, sharded.u32_add_owners(1)).to_bytes() == want
    hw = system.host_witness(traces, packed)
    assert system.prove_sharded(hw, comm, sharded.u32_add_owners(1)).to_bytes() == want
    assert comm.bytes_moved > 0
    comm.close()
    print("rccl-self ok", log_adds, flush=True)
"""


@pytest.mark.parametrize("var", ["", "MSAMD_SHARDED_HOST_SYNC", "MSAMD_SHARDED_PACK", "MSAMD_SHARDED_GENERAL"])
def test_native_rccl_transport_real_calls_to_self(var):
    """MSAMD_RCCL_SELF=1: the single rank creates a real communicator (ncclCommInitRank) and every exchange of the joint prover
    goes through ncclSend / ncclRecv to ITSELF and ncclAllGather - the only way to execute the RCCL transport's calls (dlsym'd
    signatures, groups of up to 128 operations, stream / event ordering around RCCL's kernels) on a box with one GPU, where
    RCCL refuses a second rank on the same device. In a child process under a hard time limit: a hang must not reach the box's."""
    import subprocess
    import sys

    env = dict(os.environ, MSAMD_RCCL_SELF="1", MSAMD_SHARDED_WORLD1_TRANSPORT="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if var:
        env[var] = "1"
    r = subprocess.run([sys.executable, "-c", _RCCL_SELF_SCRIPT, ROOT], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and r.stdout.count("rccl-self ok") == 3, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


def test_native_rccl_unique_id(pkg):
    import importlib

    sharded = importlib.import_module("multi_stark_amd.sharded")
    a, b = sharded.RcclComm.unique_id(), sharded.RcclComm.unique_id()
    assert len(a) == 128 and a != b


# ---------------------------------------------------------------------------------------------------------------------
# Thread ranks over the library's in-process transport: world 8 (BASELINE config 3: [ByteTable, U32Add x 8], one adder per
# rank, benches/multi_stark.rs:260-267 x 8) on the one GPU, in ONE process (eight processes on the card would trip the box's
# process guard; RCCL refuses two ranks on one device). log2 N = 3 head rounds of FRI, cap_height up to 3, 7-peer exchanges.
def _params_for(fe, variant):
    if variant.startswith("cap"):  # commitments are 2^k-digest caps, 4-coefficient final polynomial, host-driven FRI rounds
        return fe.Params(log_blowup=2, cap_height=int(variant[3:]), log_final_poly_len=2, num_queries=20, commit_proof_of_work_bits=3,
                         query_proof_of_work_bits=5)
    if variant.startswith("arity"):  # FRI rounds of arity 2^k (max_log_arity = k): no row-sharded FRI head, host-driven rounds
        return fe.Params(log_blowup=2, max_log_arity=int(variant[5:]), num_queries=20, commit_proof_of_work_bits=3, query_proof_of_work_bits=5)
    return fe.bench_params()


def _thread_rank(pkg, fe, sharded, oracle, rank, group, log_adds, variant, shared):
    world = group.world
    ctx = pkg.Context(0)
    system = pkg.System.new(ctx, _params_for(fe, variant), fe.multi_u32_add_system_inputs(world))
    traces, packed = shared["traces"], shared["packed"]
    owners = sharded.u32_add_owners(world)
    # (a rank's buffers are its own, as they would be in a process of its own: the host-resident witness page-locks them)
    mine = [t.copy() if owners[i] in (-1, rank) else None for i, t in enumerate(traces)]
    remote = {i: traces[i].shape[0] for i in range(len(traces)) if owners[i] not in (-1, rank)}
    w = system.witness(mine, packed, remote_heights=remote)
    comm = group.comm(ctx, rank)
    try:
        proof = system.prove_sharded(w, comm, owners).to_bytes()
        hw = system.host_witness(mine, packed, remote_heights=remote)  # traces + this rank's claim slice uploaded inside the proof
        for _ in range(2):
            assert system.prove_sharded(hw, comm, owners).to_bytes() == proof, "host-resident sharded proof differs"
        del hw
        again = system.prove_sharded(w, comm, owners, want_times=True)
        assert again.to_bytes() == proof and again.stage_ms["total"] > 0
        if rank == 0:
            full = system.witness(traces, packed)
            want = system.prove_multiple_claims(full).to_bytes()
            assert proof == want, "sharded proof differs from the single-GPU proof"
            osys = oracle.System(system.blob)
            assert osys.verify(packed, proof) == 0
            if log_adds <= 9:
                assert osys.prove(traces, packed) == proof, "sharded proof differs from the oracle's proof"
            assert system.verify_multiple_claims(packed, proof) == 0
        assert world == 1 or comm.bytes_moved > 0
        if world > 1:
            # the library's record of its calls into the transport (what a watchdog prints when a peer never arrives)
            text, seq, inside = ctx.comm_progress()
            assert seq > 10 and not inside and "rank %d of %d" % (rank, world) in text and "FRI queries" in text, (text, seq, inside)
        return hashlib.sha256(proof).hexdigest()
    finally:
        comm.close()


@pytest.mark.parametrize("world,log_adds,variant", [
    (8, 8, "bench"), (8, 10, "bench"), (8, 8, "cap3"), (8, 10, "cap3"), (8, 9, "cap1"), (8, 9, "pack"), (8, 9, "host-sync"),
    (8, 9, "no-overlap"), (8, 9, "full-fri"), (2, 9, "bench"), (4, 10, "bench"), (4, 9, "pack"), (1, 9, "bench"), (4, 9, "arity3"), (8, 8, "arity2"),
    # caps TALLER than the ranks' sub-trees are apart (cap_height > log2 ranks: the cap is made of layers inside the sub-trees),
    # up to a cap that is the byte table's whole leaf layer and beyond the device transcript's one-chunk limit
    (8, 9, "cap4"), (8, 13, "cap4"), (4, 9, "cap5"), (2, 9, "cap3"), (1, 9, "cap2"), (1, 13, "cap2"), (4, 8, "cap12")])
def test_thread_ranks_proof_equals_single_gpu_proof(pkg, fe, oracle, world, log_adds, variant, monkeypatch):
    import importlib

    env = {"pack": "MSAMD_SHARDED_PACK", "host-sync": "MSAMD_SHARDED_HOST_SYNC", "no-overlap": "MSAMD_SHARDED_NO_OVERLAP",
           "full-fri": "MSAMD_SHARDED_FULL_FRI"}.get(variant)
    if env:
        monkeypatch.setenv(env, "1")
    sharded = importlib.import_module("multi_stark_amd.sharded")
    traces, claims = fe.multi_u32_add_witness(world, 1 << log_adds)
    shared = {"traces": traces, "packed": fe.pack_claims(claims)}
    group = sharded.LocalGroup(world)
    try:
        res = group.run(lambda rank, g: _thread_rank(pkg, fe, sharded, oracle, rank, g, log_adds, variant, shared))
    finally:
        group.close()
    assert len(set(res)) == 1, res  # every rank holds the same proof bytes


def test_config3_full_size_eight_thread_ranks(pkg, fe):
    """BASELINE config 3 AS SPECIFIED and at FULL size - [ByteTable, U32Add x 8], 2^20 additions per rank, 8.4 M claims,
    host-resident witnesses, eight ranks - on the one GPU of the test box (thread ranks time-share it): every rank's bytes
    equal the single-GPU proof of the same nine-circuit system, and the library's verifier accepts (tools/config3_one_gpu.py
    is the timed form of the same run)."""
    import importlib

    sharded = importlib.import_module("multi_stark_amd.sharded")
    world, log_adds = 8, 20
    traces, claims = fe.multi_u32_add_witness(world, 1 << log_adds)
    packed = fe.pack_claims(claims)
    owners = sharded.u32_add_owners(world)
    ctx0 = pkg.Context(0)
    sys0 = pkg.System.new(ctx0, fe.bench_params(), fe.multi_u32_add_system_inputs(world))
    full = sys0.witness(traces, packed)
    want = sys0.prove_multiple_claims(full).to_bytes()
    assert sys0.verify_multiple_claims(packed, want) == 0
    del full
    ctx0.trim()

    def body(rank, group):
        ctx = pkg.Context(0)
        system = pkg.System.new(ctx, fe.bench_params(), fe.multi_u32_add_system_inputs(world))
        mine = [t if owners[i] in (-1, rank) else None for i, t in enumerate(traces)]
        remote = {i: traces[i].shape[0] for i in range(len(traces)) if owners[i] not in (-1, rank)}
        w = system.host_witness(mine, packed, remote_heights=remote)
        comm = group.comm(ctx, rank)
        try:
            got = [system.prove_sharded(w, comm, owners).to_bytes() for _ in range(2)]
            return got[0] == want and got[1] == want, comm.bytes_moved
        finally:
            comm.close()
            del w
            ctx.trim()

    group = sharded.LocalGroup(world)
    try:
        res = group.run(body)
    finally:
        group.close()
    assert all(ok for ok, _ in res), [ok for ok, _ in res]
    # 7/8 of each rank's stage-1 and stage-2 LDEs (14 + 26 columns x 2^22 rows) leave it per proof, plus the small gathers
    assert res[0][1] / 2 > 0.875 * 40 * (1 << 22) * 8


def test_joint_prover_host_synchronisations(pkg, fe):
    """the joint prover keeps the transcript on the device like the one-GPU prover (the commitments' top levels are hashed there
    too, the query openings of the input commitments are gathered and exchanged behind the device-sampled indices): per proof
    at the bench shape it waits for the device exactly as often as ms_prove does, at one rank and at four"""
    import importlib

    sharded = importlib.import_module("multi_stark_amd.sharded")
    log_adds = 13
    for world in (1, 4):
        traces, claims = fe.multi_u32_add_witness(world, 1 << log_adds)
        packed = fe.pack_claims(claims)
        owners = sharded.u32_add_owners(world)
        ctx0 = pkg.Context(0)
        sys0 = pkg.System.new(ctx0, fe.bench_params(), fe.multi_u32_add_system_inputs(world))
        w0 = sys0.host_witness(traces, packed)
        want = sys0.prove_multiple_claims(w0).to_bytes()
        a = ctx0.sync_count()
        assert sys0.prove_multiple_claims(w0).to_bytes() == want
        plain = ctx0.sync_count() - a
        assert plain == 2, plain  # the opened values, FRI

        def body(rank, group):
            ctx = pkg.Context(0)
            system = pkg.System.new(ctx, fe.bench_params(), fe.multi_u32_add_system_inputs(world))
            mine = [t if owners[i] in (-1, rank) else None for i, t in enumerate(traces)]
            remote = {i: traces[i].shape[0] for i in range(len(traces)) if owners[i] not in (-1, rank)}
            w = system.host_witness(mine, packed, remote_heights=remote)
            comm = group.comm(ctx, rank)
            try:
                assert system.prove_sharded(w, comm, owners).to_bytes() == want
                b = ctx.sync_count()
                assert system.prove_sharded(w, comm, owners).to_bytes() == want
                return ctx.sync_count() - b
            finally:
                comm.close()

        group = sharded.LocalGroup(world)
        try:
            counts = group.run(body)
        finally:
            group.close()
        # (the FRI rounds that run on row shards keep their transcript steps on the device too: merkle_top_challenge)
        assert max(counts) == plain, (world, counts, plain)  # exactly the one-GPU prover's two waits: opened values, FRI


@pytest.mark.parametrize("world,log_adds", [(8, 12), (4, 13), (2, 13)])
def test_claims_sliced_on_the_host(pkg, fe, world, log_adds):
    """every rank holds only the part of the claims' data it reads (ms_claims_slice_range / ms_witness_create_host_sliced: all
    offsets, its element range, the first 130 elements): the same bytes as with all claims on every rank; the parts together
    are little more than one copy of the data; a rank given too small a part fails with a reason"""
    import importlib

    import numpy as np

    sharded = importlib.import_module("multi_stark_amd.sharded")
    traces, claims = fe.multi_u32_add_witness(world, 1 << log_adds)
    offs, data = fe.pack_claims(claims)
    owners = sharded.u32_add_owners(world)
    ctx0 = pkg.Context(0)
    sys0 = pkg.System.new(ctx0, fe.bench_params(), fe.multi_u32_add_system_inputs(world))
    want = sys0.prove_multiple_claims(sys0.witness(traces, (offs, data))).to_bytes()
    heights = [t.shape[0] for t in traces]
    held = []

    def body(rank, group):
        ctx = pkg.Context(0)
        system = pkg.System.new(ctx, fe.bench_params(), fe.multi_u32_add_system_inputs(world))
        first, count = system.claims_slice_range(heights, offs, rank, world)
        held.append(count)
        mine = [t if owners[i] in (-1, rank) else None for i, t in enumerate(traces)]
        remote = {i: traces[i].shape[0] for i in range(len(traces)) if owners[i] not in (-1, rank)}
        w = system.host_witness_sliced(mine, offs, first, data[first:first + count].copy(), data[:130].copy(), remote_heights=remote)
        comm = group.comm(ctx, rank)
        try:
            got = [system.prove_sharded(w, comm, owners).to_bytes() for _ in range(2)]
            with pytest.raises(pkg.MstarkError):
                system.prove_multiple_claims(w)  # the one-GPU prover needs every claim
            return got[0] == want and got[1] == want
        finally:
            comm.close()

    group = sharded.LocalGroup(world)
    try:
        assert all(group.run(body))
    finally:
        group.close()
    assert sum(held) <= data.size + 16 * world * 8, (held, data.size)  # the parts overlap by a claim or two at the seams
    # a part that is too small is an error with a reason, on that rank (its peers are released by the transport's abort)
    def short_body(rank, group):
        ctx = pkg.Context(0)
        system = pkg.System.new(ctx, fe.bench_params(), fe.multi_u32_add_system_inputs(world))
        first, count = system.claims_slice_range(heights, offs, rank, world)
        if rank == world - 1:
            count -= 8
        mine = [t if owners[i] in (-1, rank) else None for i, t in enumerate(traces)]
        remote = {i: traces[i].shape[0] for i in range(len(traces)) if owners[i] not in (-1, rank)}
        w = system.host_witness_sliced(mine, offs, first, data[first:first + count].copy(), data[:130].copy(), remote_heights=remote)
        comm = group.comm(ctx, rank)
        try:
            system.prove_sharded(w, comm, owners)
        finally:
            comm.close()

    group = sharded.LocalGroup(world)
    try:
        with pytest.raises(pkg.MstarkError, match="does not hold the part of the claims"):
            group.run(short_body)
    finally:
        group.close()


def test_thread_ranks_failure_does_not_hang(pkg, fe, monkeypatch):
    """a rank that fails in the middle of a joint proof (injected allocation failure) makes every rank return an error -
    nobody waits for it forever - and the same contexts prove again afterwards"""
    import importlib

    sharded = importlib.import_module("multi_stark_amd.sharded")
    monkeypatch.setenv("MSAMD_LOCAL_TIMEOUT_S", "20")
    world = 4
    traces, claims = fe.multi_u32_add_witness(world, 1 << 9)
    packed = fe.pack_claims(claims)
    owners = sharded.u32_add_owners(world)
    ctxs = [pkg.Context(0) for _ in range(world)]
    systems = [pkg.System.new(c, fe.bench_params(), fe.multi_u32_add_system_inputs(world)) for c in ctxs]
    wits = []
    for r in range(world):
        mine = [t if owners[i] in (-1, r) else None for i, t in enumerate(traces)]
        remote = {i: traces[i].shape[0] for i in range(len(traces)) if owners[i] not in (-1, r)}
        wits.append(systems[r].witness(mine, packed, remote_heights=remote))

    def attempt(fail_rank, nth):
        group = sharded.LocalGroup(world)

        def body(rank, g):
            comm = g.comm(ctxs[rank], rank)
            try:
                if rank == fail_rank:
                    ctxs[rank].debug_fail_alloc(nth)
                return systems[rank].prove_sharded(wits[rank], comm, owners).to_bytes()
            finally:
                ctxs[rank].debug_fail_alloc(0)
                comm.close()

        try:
            return group.run(body)
        finally:
            group.close()

    good = attempt(-1, 0)
    assert len(set(good)) == 1
    for nth in (3, 25, 60):
        with pytest.raises(pkg.MstarkError):
            attempt(2, nth)
        assert attempt(-1, 0) == good


def test_thread_ranks_peer_that_never_arrives(pkg, fe, monkeypatch):
    """a peer that never joins the proof: the waiting rank gets an error that names the exchange it waited in - not a hang"""
    import importlib
    import time

    sharded = importlib.import_module("multi_stark_amd.sharded")
    monkeypatch.setenv("MSAMD_LOCAL_TIMEOUT_S", "3")
    world = 2
    traces, claims = fe.multi_u32_add_witness(world, 1 << 8)
    packed = fe.pack_claims(claims)
    owners = sharded.u32_add_owners(world)
    group = sharded.LocalGroup(world)
    seen = {}

    def body(rank, g):
        ctx = pkg.Context(0)
        comm = g.comm(ctx, rank)
        try:
            if rank == 1:
                time.sleep(6)  # never calls ms_prove_sharded
                return "absent"
            system = pkg.System.new(ctx, fe.bench_params(), fe.multi_u32_add_system_inputs(world))
            mine = [t if owners[i] in (-1, rank) else None for i, t in enumerate(traces)]
            w = system.witness(mine, packed, remote_heights={2: traces[2].shape[0]})
            t0 = time.time()
            try:
                system.prove_sharded(w, comm, owners)
            except pkg.MstarkError as e:
                seen["error"], seen["after"] = str(e), time.time() - t0
                seen["progress"] = ctx.comm_progress()
                return "failed as it should"
            return "returned?!"
        finally:
            comm.close()

    try:
        res = group.run(body)
    finally:
        group.close()
    assert res == ["failed as it should", "absent"], (res, seen)
    assert "never arrived" in seen["error"] and "stage-1 commit" in seen["error"] and seen["after"] < 5.5, seen
    assert "stage-1 commit" in seen["progress"][0]


def _thread_random_rank(pkg, fe, fz, np, rank, group, seed, n_cases):
    """the random systems of _random_worker on thread ranks (same generator, same stream on every rank)"""
    world = group.world
    ctx = pkg.Context(0)
    comm = group.comm(ctx, rank)
    try:
        master = np.random.default_rng(seed)
        done = 0
        for case in range(n_cases):
            rng = np.random.default_rng(master.integers(0, 1 << 62))
            lb = int(rng.integers(1, 3))
            lw = world.bit_length() - 1
            # (caps up to two levels taller than log2 ranks: gathered from inside the sub-trees)
            params = fe.Params(log_blowup=lb, cap_height=int(rng.integers(0, lw + 3)), log_final_poly_len=0,
                               num_queries=int(rng.integers(1, 12)), commit_proof_of_work_bits=int(rng.integers(0, 5)),
                               query_proof_of_work_bits=int(rng.integers(0, 5)))
            shard_ci, shard_w, fixed_h = fz.random_circuit(rng, fe, lb)
            shard_h = fixed_h if fixed_h else 1 << int(rng.integers(3, 10))
            if shard_h < world:
                continue
            circuits, traces, owners = [], [], []
            n_rep = int(rng.integers(0, 3))
            rep_positions = sorted(int(x) for x in rng.integers(0, world + 1, n_rep))
            k = 0
            for slot in range(world + 1):
                for _ in range(rep_positions.count(slot)):
                    ci, w, fh = fz.random_circuit(rng, fe, lb)
                    h = fh if fh else 1 << int(rng.integers(3, 10))
                    circuits.append(ci)
                    traces.append(fz.rand_field(rng, (h, w)))
                    owners.append(-1)
                if slot < world:
                    circuits.append(shard_ci)
                    traces.append(fz.rand_field(rng, (shard_h, shard_w)))
                    owners.append(k)
                    k += 1
            if any(t.shape[0] < world for t in traces):
                continue
            claims = [[int(x) for x in fz.rand_field(rng, int(rng.integers(0, 5)))] for _ in range(int(rng.integers(0, 4)))]
            if rng.random() < 0.4:  # more than 8192 claim words: the joint prover keeps the outer transcript on the device
                claims += [[int(x) for x in fz.rand_field(rng, int(rng.integers(2000, 4000)))] for _ in range(int(rng.integers(3, 6)))]
            packed = fe.pack_claims(claims)
            try:
                compiled = [fe.compile_circuit(c) for c in circuits]
            except fe.CompileError:
                continue
            try:
                system = pkg.System(ctx, fe.system_blob(params, compiled), len(compiled))
            except pkg.MstarkError:
                continue  # e.g. a random constraint above the degree bound: every rank skips alike
            mine = [t if owners[i] in (-1, rank) else None for i, t in enumerate(traces)]
            remote = {i: traces[i].shape[0] for i in range(len(traces)) if owners[i] not in (-1, rank)}
            proof = system.prove_sharded(system.witness(mine, packed, remote_heights=remote), comm, owners).to_bytes()
            if case % 3 == 0:  # and from a host-resident witness (this rank's slice of the claims uploaded inside the proof)
                assert system.prove_sharded(system.host_witness(mine, packed, remote_heights=remote), comm, owners).to_bytes() == proof
            if rank == case % world:  # the single-GPU proof of the same system, checked by a different rank every case
                want = system.prove_multiple_claims(system.witness(traces, packed)).to_bytes()
                assert proof == want, "random case %d: sharded proof differs from the single-GPU proof" % case
            done += 1
        return done
    finally:
        comm.close()


@pytest.mark.parametrize("world,seed", [(8, 11), (4, 12)])
def test_thread_ranks_random_systems(pkg, fe, world, seed, monkeypatch):
    import importlib

    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity as fz

    monkeypatch.setenv("MSAMD_NO_JIT", "1")
    sharded = importlib.import_module("multi_stark_amd.sharded")
    n_cases = int(os.environ.get("MSAMD_SHARDED_FUZZ_CASES", "30"))
    group = sharded.LocalGroup(world)
    try:
        seed += 1000 * int(os.environ.get("MSAMD_SHARDED_FUZZ_SEED", "0"))  # (other systems for an ad-hoc soak)
        res = group.run(lambda rank, g: _thread_random_rank(pkg, fe, fz, np, rank, g, seed, n_cases))
    finally:
        group.close()
    assert len(set(res)) == 1 and res[0] >= 10, res
    print("thread ranks, random systems:", res[0], "cases, world", world)


# ---------------------------------------------------------------------------------------------------------------------
# General ownership (ms_comm.scatter_cols_start): any number of circuits per rank, any shapes, replicated tables.
def _wide_and_tables(fe, np, W=96, H=64):
    """the structure of the reference's largest scenario (src/test_circuits/blake3.rs:2215-2613: one wide circuit beside tables
    of other widths and heights, linked by lookups), scaled down: a W-column product chain pushes two pairs that two tables of
    other heights pull, next to a byte-range table and a second chain of another height"""
    P = fe.P
    rng = np.random.default_rng(177)

    def chain(w, h, seed):
        r = np.random.default_rng(seed)
        m = np.zeros((h, w), dtype=object)
        base = int(r.integers(1, 1 << 20))
        m[:, 0] = [base + i for i in range(h)]
        m[:, 1] = [int(x) for x in r.integers(1, 1 << 20, h)]
        for i in range(2, w):
            m[:, i] = (m[:, i - 2] * m[:, i - 1]) % P
        return m.astype(np.uint64)

    def chain_air(w, pushes):
        def ev(b):
            local, nxt = b.main()
            for i in range(2, w):
                b.assert_eq(local[i - 2] * local[i - 1], local[i])
            b.when_transition().assert_eq(nxt[0], local[0] + fe.Expr.const(1))

        return fe.lookup_air(w, ev, pushes)

    var, one = fe.Expr.main, fe.Expr.const(1)
    wide = chain(W, H, 1)
    other = chain(12, 4 * H, 2)
    wide_air = chain_air(W, [fe.Lookup.push(one, [var(0), var(1)]), fe.Lookup.push(one, [var(W - 2), var(W - 1)])])
    other_air = chain_air(12, [fe.Lookup.push(one, [var(3), var(4)])])
    t1 = np.stack([wide[:, 0], wide[:, 1]], axis=1)
    t2 = np.zeros((2 * H, 3), dtype=np.uint64)
    t2[:H, 0] = 1
    t2[:H, 1] = wide[:, W - 2]
    t2[:H, 2] = wide[:, W - 1]
    t3 = np.stack([other[:, 3], other[:, 4]], axis=1)
    a1 = fe.lookup_air(2, None, [fe.Lookup.pull(one, [var(0), var(1)])])
    a2 = fe.lookup_air(3, lambda b: b.assert_bool(b.main()[0][0]), [fe.Lookup.pull(var(0), [var(1), var(2)])])
    a3 = fe.lookup_air(2, None, [fe.Lookup.pull(one, [var(0), var(1)])])
    del rng
    return [a1, wide_air, a2, other_air, a3], [t1, wide, t2, other, t3]


GENERAL_MAPS = {  # owners of [table1, wide, table2, other chain, table3]
    2: [[-1, 0, -1, 1, -1], [1, 0, 1, 0, -1], [-1, 1, 1, 1, 0], [0, 0, 0, 0, 0]],
    4: [[-1, 0, -1, 3, -1], [2, 0, 2, 1, 3], [-1, 1, -1, 1, -1]],
    8: [[-1, 5, 2, 0, -1]],
}


def _general_rank(pkg, fe, sharded, oracle, np, rank, group, maps, shared):
    ctx = pkg.Context(0)
    comm = group.comm(ctx, rank)
    inputs, traces = shared["inputs"], shared["traces"]
    packed = fe.pack_claims([])
    out = []
    try:
        for params in (fe.test_params(), fe.Params(log_blowup=2, cap_height=1, log_final_poly_len=1, num_queries=9, commit_proof_of_work_bits=2,
                                                   query_proof_of_work_bits=3)):
            system = pkg.System.new(ctx, params, inputs)
            for owners in maps:
                mine = [t.copy() if owners[i] in (-1, rank) else None for i, t in enumerate(traces)]
                remote = {i: traces[i].shape[0] for i in range(len(traces)) if owners[i] not in (-1, rank)}
                proof = system.prove_sharded(system.witness(mine, packed, remote_heights=remote), comm, owners).to_bytes()
                hproof = system.prove_sharded(system.host_witness(mine, packed, remote_heights=remote), comm, owners).to_bytes()
                assert hproof == proof, "host-resident witness: joint proof differs"
                if rank == 0:
                    want = system.prove_multiple_claims(system.witness(traces, packed)).to_bytes()
                    assert proof == want, "owners %s: joint proof differs from the single-GPU proof" % (owners,)
                    osys = oracle.System(system.blob)
                    assert osys.verify(packed, proof) == 0 and osys.prove(traces, packed) == proof
                out.append(hashlib.sha256(proof).hexdigest())
        return out
    finally:
        comm.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_general_ownership_wide_circuit_and_tables(pkg, fe, oracle, world):
    """circuits of different widths and heights, several (or none) per rank, replicated tables: every map gives the bytes of
    the single-GPU proof and of the oracle's"""
    import importlib

    import numpy as np

    sharded = importlib.import_module("multi_stark_amd.sharded")
    inputs, traces = _wide_and_tables(fe, np)
    shared = {"inputs": inputs, "traces": traces}
    group = sharded.LocalGroup(world)
    try:
        res = group.run(lambda rank, g: _general_rank(pkg, fe, sharded, oracle, np, rank, g, GENERAL_MAPS[world], shared))
    finally:
        group.close()
    assert all(r == res[0] for r in res), res
    assert len(set(res[0])) == 2  # one proof per parameter set, whatever the map


def test_general_pattern_on_the_uniform_system(pkg, fe, oracle, monkeypatch):
    """BASELINE config 3's layout through the per-matrix hand-out (MSAMD_SHARDED_GENERAL=1): same bytes as the all-to-all"""
    import importlib

    sharded = importlib.import_module("multi_stark_amd.sharded")
    world, log_adds = 4, 9
    traces, claims = fe.multi_u32_add_witness(world, 1 << log_adds)
    shared = {"traces": traces, "packed": fe.pack_claims(claims)}
    res = {}
    for mode in ("uniform", "general"):
        if mode == "general":
            monkeypatch.setenv("MSAMD_SHARDED_GENERAL", "1")
        group = sharded.LocalGroup(world)
        try:
            res[mode] = group.run(lambda rank, g: _thread_rank(pkg, fe, sharded, oracle, rank, g, log_adds, "bench", shared))
        finally:
            group.close()
    assert set(res["uniform"]) == set(res["general"]) and len(set(res["general"])) == 1


def _general_random_rank(pkg, fe, fz, np, rank, group, seed, n_cases):
    """random systems with random owner maps: 1-5 circuits of random shapes, each replicated or owned by a random rank"""
    world = group.world
    ctx = pkg.Context(0)
    comm = group.comm(ctx, rank)
    try:
        master = np.random.default_rng(seed)
        done = 0
        for case in range(n_cases):
            rng = np.random.default_rng(master.integers(0, 1 << 62))
            lb = int(rng.integers(1, 3))
            lw = world.bit_length() - 1
            params = fe.Params(log_blowup=lb, cap_height=int(rng.integers(0, lw + 3)), log_final_poly_len=0,
                               num_queries=int(rng.integers(1, 12)), commit_proof_of_work_bits=int(rng.integers(0, 5)),
                               query_proof_of_work_bits=int(rng.integers(0, 5)))
            circuits, traces, owners = [], [], []
            for _ in range(int(rng.integers(1, 6))):
                ci, w, fh = fz.random_circuit(rng, fe, lb)
                h = fh if fh else 1 << int(rng.integers(3, 10))
                circuits.append(ci)
                traces.append(fz.rand_field(rng, (h, w)))
                owners.append(int(rng.integers(-1, world)))
            if any(t.shape[0] < world for t in traces):
                continue
            claims = [[int(x) for x in fz.rand_field(rng, int(rng.integers(0, 5)))] for _ in range(int(rng.integers(0, 4)))]
            if rng.random() < 0.4:  # more than 8192 claim words: the outer transcript on the device
                claims += [[int(x) for x in fz.rand_field(rng, int(rng.integers(2000, 4000)))] for _ in range(int(rng.integers(3, 6)))]
            packed = fe.pack_claims(claims)
            try:
                compiled = [fe.compile_circuit(c) for c in circuits]
            except fe.CompileError:
                continue
            try:
                system = pkg.System(ctx, fe.system_blob(params, compiled), len(compiled))
            except pkg.MstarkError:
                continue
            mine = [t if owners[i] in (-1, rank) else None for i, t in enumerate(traces)]
            remote = {i: traces[i].shape[0] for i in range(len(traces)) if owners[i] not in (-1, rank)}
            proof = system.prove_sharded(system.witness(mine, packed, remote_heights=remote), comm, owners).to_bytes()
            if rank == case % world:
                want = system.prove_multiple_claims(system.witness(traces, packed)).to_bytes()
                assert proof == want, "random case %d (owners %s): joint proof differs from the single-GPU proof" % (case, owners)
            done += 1
        return done
    finally:
        comm.close()


@pytest.mark.parametrize("world,seed", [(2, 21), (4, 22), (8, 23)])
def test_general_ownership_random_systems(pkg, fe, world, seed, monkeypatch):
    import importlib

    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity as fz

    monkeypatch.setenv("MSAMD_NO_JIT", "1")
    sharded = importlib.import_module("multi_stark_amd.sharded")
    n_cases = int(os.environ.get("MSAMD_SHARDED_FUZZ_CASES", "30"))
    group = sharded.LocalGroup(world)
    try:
        seed += 1000 * int(os.environ.get("MSAMD_SHARDED_FUZZ_SEED", "0"))
        res = group.run(lambda rank, g: _general_random_rank(pkg, fe, fz, np, rank, g, seed, n_cases))
    finally:
        group.close()
    assert len(set(res)) == 1 and res[0] >= 10, res
    print("general ownership, random systems:", res[0], "cases, world", world)


def _general_gloo_worker(rank, world, port, q):
    """the same maps over torch.distributed / gloo processes (TorchComm's staged scatter)"""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="4", OMP_WAIT_POLICY="passive")
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import importlib

        import numpy as np
        import torch.distributed as dist
        from __graft_entry__ import load_package

        pkg = load_package()
        fe = pkg.frontend
        sharded = importlib.import_module("multi_stark_amd.sharded")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        try:
            ctx = pkg.Context(0)
            comm = sharded.TorchComm(0)
            inputs, traces = _wide_and_tables(fe, np)
            packed = fe.pack_claims([])
            system = pkg.System.new(ctx, fe.test_params(), inputs)
            shas = []
            for owners in GENERAL_MAPS[world]:
                mine = [t if owners[i] in (-1, rank) else None for i, t in enumerate(traces)]
                remote = {i: traces[i].shape[0] for i in range(len(traces)) if owners[i] not in (-1, rank)}
                proof = system.prove_sharded(system.witness(mine, packed, remote_heights=remote), comm, owners).to_bytes()
                if rank == 0:
                    assert proof == system.prove_multiple_claims(system.witness(traces, packed)).to_bytes()
                shas.append(hashlib.sha256(proof).hexdigest())
            q.put((rank, ",".join(shas), comm.bytes_moved))
        finally:
            dist.barrier()
            dist.destroy_process_group()
    except BaseException as e:
        q.put((rank, "ERROR: %r" % (e,), 0))
        raise


def test_general_ownership_over_gloo_processes():
    world = 2
    port = 29800 + (os.getpid() % 1000)
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_general_gloo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
    assert all(not str(r[1]).startswith("ERROR") for r in res), res
    assert all(p.exitcode == 0 for p in procs)
    assert len({r[1] for r in res}) == 1
