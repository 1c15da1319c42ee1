"""Stream-ordering regression net. The prover runs on up to four streams at once - main, side (short circuits beside long
ones, the claims digest beside the hashing), copy (a host-resident witness's uploads) and, in a joint proof, the transport's -
and a consumer that forgets to wait for its producer reads what the PREVIOUS proof left in the buffer: right bytes for the same
witness, wrong ones for another. Every case therefore alternates TWO witnesses and runs under each delay diagnostic, which
holds one stream back at every fork / upload so that a missing wait fails every time instead of once in a while:
  MSAMD_SIDE_DELAY_US  the side stream starts late at every fork
  MSAMD_MAIN_DELAY_US  the main stream is held back behind every fork (the side stream runs ahead)
  MSAMD_COPY_DELAY_US  the copy stream starts late at every proof's upload
(the race fixed in round 3 - a long read-back issued on the main stream without waiting for the side stream - is of this kind;
tests/test_blake3_circuit.py::test_side_stream_results_are_awaited keeps the system that found it)."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DELAYS = ["none", "MSAMD_SIDE_DELAY_US", "MSAMD_MAIN_DELAY_US", "MSAMD_COPY_DELAY_US"]


def _set_delay(monkeypatch, which):
    if which != "none":
        monkeypatch.setenv(which, "1500")


def _alternate(system, wits, want):
    """device- and host-resident witnesses of two inputs in turn: a stale buffer must never pass for this proof's values"""
    for k in (0, 1, 0, 3, 2, 1, 2):
        assert system.prove_multiple_claims(wits[k]).to_bytes() == want[k % 2], k


@pytest.mark.parametrize("which", DELAYS)
def test_bench_workload_two_witnesses(pkg, ctx, fe, oracle, monkeypatch, which):
    _set_delay(monkeypatch, which)
    g = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    o = oracle.System(g.blob)
    sets = []
    for a0, b0 in ((0xDEADBEEF, 0xCAFEBABE), (0x12345678, 0x9ABCDEF1)):
        tr, cl = fe.u32_add_bench_witness(1 << 14, a0, b0)
        sets.append((tr, fe.pack_claims(cl)))
    want = [o.prove(tr, pk) for tr, pk in sets]
    assert want[0] != want[1]
    wits = [g.witness(tr, pk) for tr, pk in sets] + [g.host_witness(tr, pk) for tr, pk in sets]
    _alternate(g, wits, want)


@pytest.mark.parametrize("which", DELAYS)
def test_nine_circuit_system_two_witnesses(pkg, ctx, fe, oracle, monkeypatch, which):
    _set_delay(monkeypatch, which)
    g = pkg.System.new(ctx, fe.bench_params(), fe.multi_u32_add_system_inputs(8))
    o = oracle.System(g.blob)
    sets = []
    for log_adds in (12, 11):  # (different heights too: buffers of another size in the pool)
        tr, cl = fe.multi_u32_add_witness(8, 1 << log_adds)
        sets.append((tr, fe.pack_claims(cl)))
    want = [o.prove(tr, pk) for tr, pk in sets]
    wits = [g.witness(tr, pk) for tr, pk in sets] + [g.host_witness(tr, pk) for tr, pk in sets]
    _alternate(g, wits, want)


@pytest.mark.parametrize("which", DELAYS)
def test_preprocessed_and_lookup_system_two_witnesses(pkg, ctx, fe, oracle, monkeypatch, which):
    """ByteCS (src/test_circuits/byte_operations.rs: a 2^16-row preprocessed table, four pull lookups) with claims"""
    _set_delay(monkeypatch, which)
    g = pkg.System.new(ctx, fe.test_params(), fe.byte_operations_inputs())
    o = oracle.System(g.blob)
    sets = []
    for seed in (5, 6):
        rng = np.random.default_rng(seed)
        calls = [(int(rng.integers(0, 4)), int(rng.integers(0, 256)), int(rng.integers(0, 256))) for _ in range(3000)]
        tr, cl = fe.byte_operations_witness(calls)
        sets.append((tr, fe.pack_claims(cl)))
    want = [o.prove(tr, pk) for tr, pk in sets]
    assert want[0] != want[1]
    wits = [g.witness(tr, pk) for tr, pk in sets] + [g.host_witness(tr, pk) for tr, pk in sets]
    _alternate(g, wits, want)


def _joint(pkg, fe, world, inputs, params, sets, owners, want):
    sharded = importlib.import_module("multi_stark_amd.sharded")

    def body(rank, group):
        ctx = pkg.Context(0)
        system = pkg.System.new(ctx, params, inputs)
        comm = group.comm(ctx, rank)
        try:
            wits = []
            for make in (system.witness, system.host_witness):
                for tr, pk in sets:
                    mine = [t if owners[i] in (-1, rank) else None for i, t in enumerate(tr)]
                    remote = {i: tr[i].shape[0] for i in range(len(tr)) if owners[i] not in (-1, rank)}
                    wits.append(make(mine, pk, remote_heights=remote))
            for k in (0, 1, 0, 3, 2, 1):
                assert system.prove_sharded(wits[k], comm, owners).to_bytes() == want[k % 2], (rank, k)
            return True
        finally:
            comm.close()

    group = sharded.LocalGroup(world)
    try:
        assert all(group.run(body))
    finally:
        group.close()


@pytest.mark.parametrize("which", DELAYS)
def test_joint_proof_four_thread_ranks_uniform(pkg, ctx, fe, oracle, monkeypatch, which):
    """BASELINE config 3's layout at four ranks (device transcript, claims digest gathered on the side stream, row-sharded FRI head)"""
    _set_delay(monkeypatch, which)
    sharded = importlib.import_module("multi_stark_amd.sharded")
    world = 4
    inputs, params = fe.multi_u32_add_system_inputs(world), fe.bench_params()
    g = pkg.System.new(ctx, params, inputs)
    sets = []
    for log_adds in (13, 12):  # (2^13: above the device-transcript threshold of 8192 claim words)
        tr, cl = fe.multi_u32_add_witness(world, 1 << log_adds)
        sets.append((tr, fe.pack_claims(cl)))
    want = [g.prove_multiple_claims(g.witness(tr, pk)).to_bytes() for tr, pk in sets]
    assert oracle.System(g.blob).verify(sets[0][1], want[0]) == 0
    _joint(pkg, fe, world, inputs, params, sets, sharded.u32_add_owners(world), want)


@pytest.mark.parametrize("which", DELAYS)
def test_joint_proof_four_thread_ranks_general(pkg, ctx, fe, oracle, monkeypatch, which):
    """general ownership (a wide circuit beside tables of other shapes, several circuits on one rank, replicated tables)"""
    _set_delay(monkeypatch, which)
    tg = importlib.import_module("test_gpu_sharded")
    world = 4
    inputs, traces = tg._wide_and_tables(fe, np)
    params = fe.Params(log_blowup=2, cap_height=1, log_final_poly_len=1, num_queries=9, commit_proof_of_work_bits=2, query_proof_of_work_bits=3)
    g = pkg.System.new(ctx, params, inputs)
    none = fe.pack_claims([])
    # (one input here - its traces are tied together by lookups - proved from device- and host-resident witnesses in turn)
    sets = [(traces, none), (traces, none)]
    want = [g.prove_multiple_claims(g.witness(traces, none)).to_bytes()] * 2
    assert oracle.System(g.blob).verify(none, want[0]) == 0
    _joint(pkg, fe, world, inputs, params, sets, tg.GENERAL_MAPS[world][1], want)


@pytest.mark.parametrize("which", DELAYS)
def test_babybear_two_witnesses(pkg, ctx, fe, monkeypatch, which):
    """the second configuration keeps to one stream plus the copy stream of its host-resident witness"""
    import oracle_bb

    _set_delay(monkeypatch, which)
    bb = pkg.babybear
    with fe.field(fe.BABYBEAR):
        g = bb.System.new(ctx, fe.test_params(), fe.mul_air_inputs(), fe.poseidon2_constants())
        none = fe.pack_claims([])
        trs = [fe.mul_air_trace(1 << 12), fe.mul_air_trace(1 << 13)]
    o = oracle_bb.System(g.blob)
    want = [o.prove([t], none) for t in trs]
    wits = [g.witness([t], none) for t in trs] + [g.host_witness([t], none) for t in trs]
    _alternate(g, wits, want)


# Random systems under each delay, device- and host-resident witnesses alternating over numpy buffers whose addresses recur
# (tools/fuzz_parity.py FUZZ_BIG FUZZ_PARAMS, a process of its own per delay): the run that ended with a GPU memory access fault
# on a host address in round 4 (asynchronous copies from pageable caller memory beside hipHostRegister / hipHostUnregister of the
# same ranges) - the first thirty cases of that seed take a few seconds per delay
@pytest.mark.parametrize("which", DELAYS[1:])
def test_random_systems_with_alternating_witnesses_under_delays(which):
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FUZZ_BIG="1", FUZZ_PARAMS="1", MSAMD_NO_JIT="1")
    env[which] = "300"
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "30", "5020"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert r.stdout.strip().splitlines()[-1].startswith("OK: 30 random systems"), r.stdout[-500:]
