"""The reference's largest scenario - the BLAKE3 compression system of /root/reference/src/test_circuits/blake3.rs (nine circuits
linked by lookups, a 2^16-row preprocessed table, a 2625-column circuit with 73 lookups) - authored in the front-end
(multi-stark_amd/blake3_circuit.py) and proved: CPU tests on the oracle, `-m gpu` tests byte-identical through the C ABI.
Covers `g_function_test_vector`, `compression_test_vector` (:2616-2746), `test_compression_reference_compatibility` (:2215-2340)
and `test_all_claims` (:2343-2613)."""
import importlib

import numpy as np
import pytest


@pytest.fixture(scope="module")
def b3(pkg):
    return importlib.import_module("multi_stark_amd.blake3_circuit")


@pytest.fixture(scope="module")
def compiled(pkg, fe, b3):
    inputs = b3.blake3_system_inputs()
    comp = [fe.compile_circuit(ci) for ci in inputs]
    return inputs, comp, fe.system_blob(fe.test_params(), comp)  # the tests' parameters: blowup 2, 64 queries, no proof of work


def test_reference_test_vectors(b3):
    t = b3._g(0x11111111, 0x22222222, 0x33333333, 0x44444444, 0x55555555, 0x66666666)  # :2616-2644
    assert (t[8], t[13], t[11], t[10]) == (0xCCCCCCCB, 0x45B64444, 0x06FFFFFF, 0x07000000)
    name, claims = b3.all_claims_cases()[-1]  # :2646-2746 (the witness generator checks state_out against its own rounds)
    assert name == "compression" and b3.blake3_witness(claims)[8].shape == (1, 2625)


def test_hasher_against_the_oracles_blake3(b3, oracle):
    for n in (0, 1, 63, 64, 65, 1023, 1024, 1025, 2048, 2049, 3000, 5000, 8192):
        data = bytes((i * 7 + 3) & 255 for i in range(n))
        infos, dig = b3.blake3_compressions(data)
        assert dig == oracle.hash_bytes(data), n
        blocks = max(1, (n + 63) // 64)
        chunks = max(1, (n + 1023) // 1024)
        assert len(infos) == blocks + chunks - 1  # every block once, every parent node once
        for info in infos:
            assert b3.compress(info["cv"], info["block_words"], info["counter_low"] | info["counter_high"] << 32, info["block_len"],
                               info["flags"]) == info["output"]


def test_system_shape(fe, b3, compiled, oracle):
    inputs, comp, blob = compiled
    s = oracle.System(blob)
    want = [(2, 3, 65536, 2), (13, 0, 0, 5), (14, 0, 0, 9), (9, 0, 0, 3), (9, 0, 0, 3), (25, 0, 0, 1), (25, 0, 0, 1), (81, 0, 0, 15), (2625, 0, 0, 73)]
    for ci, (w, pw, ph, nl) in enumerate(want):
        info = s.circuit_info(ci)
        assert (info["main_width"], info["pre_width"], info["pre_height"], info["num_lookups"]) == (w, pw, ph, nl), ci
    # 56 x 6 + 8 x 6 constraints of the AIR plus two per lookup column pair (src/system.rs:115-203)
    assert s.circuit_info(8)["constraint_count"] == 56 * 6 + 8 * 6 + 2 * 73 and s.circuit_info(8)["stage2_width"] == 146
    assert s.circuit_info(2)["constraint_count"] == 2 + 2 * 9 and s.circuit_info(5)["constraint_count"] == 2 + 2


def _cases(b3):
    out = list(b3.all_claims_cases())
    infos, _ = b3.blake3_compressions(bytes([0x54] * 64))  # test_compression_reference_compatibility: one compression
    assert len(infos) == 1
    out.append(("reference_compatibility", [b3.compression_claim(infos[0])]))
    return out


@pytest.mark.parametrize("idx", range(7))
def test_claims_prove_and_verify_on_the_oracle(fe, b3, compiled, oracle, idx):
    name, claims = _cases(b3)[idx]
    s = oracle.System(compiled[2])
    traces = b3.blake3_witness(claims)
    packed = fe.pack_claims(claims)
    proof = s.prove(traces, packed)
    assert s.verify(packed, proof) == 0, name
    # another claim of the same shape is not what was proved
    wrong = [list(c) for c in claims]
    wrong[0][-1] ^= 1
    assert s.verify(fe.pack_claims(wrong), proof) != 0
    # a trace that breaks a lookup balance (one byte pair less than was sent) gives a proof the verifier refuses
    traces[0][0, 1] += 1
    assert s.verify(packed, s.prove(traces, packed)) != 0


def test_wrong_claims_are_refused_by_the_witness_generator(b3):
    with pytest.raises(ValueError):
        b3.blake3_witness([[b3.U32_XOR, 1, 2, 4]])
    with pytest.raises(ValueError):
        b3.blake3_witness([[b3.ROT7, 1, 2]])
    with pytest.raises(ValueError):
        b3.blake3_witness([[b3.U32_ADD, 1, 2]])


def test_whole_hash_as_compression_claims(fe, b3, compiled, oracle):
    """every compression of a 3000-byte hash (47 blocks + 2 parents) as claims of the compression circuit: 64-row traces of 2625 columns"""
    infos, dig = b3.blake3_compressions(bytes(range(256)) * 11 + bytes(184))
    claims = [b3.compression_claim(i) for i in infos]
    traces = b3.blake3_witness(claims)
    assert traces[8].shape == (64, 2625) and traces[7].shape == (4096, 81)
    s = oracle.System(compiled[2])
    packed = fe.pack_claims(claims)
    proof = s.prove(traces, packed)
    assert s.verify(packed, proof) == 0
    assert s.verify(fe.pack_claims(claims[:-1]), proof) != 0


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(7))
def test_gpu_proofs_equal_the_oracles(pkg, ctx, fe, b3, compiled, oracle, idx):
    name, claims = _cases(b3)[idx]
    g = pkg.System.new(ctx, fe.test_params(), compiled[0])
    assert g.blob == compiled[2]
    o = oracle.System(g.blob)
    assert g.preprocessed_commit() == o.preprocessed_commit()
    traces = b3.blake3_witness(claims)
    packed = fe.pack_claims(claims)
    proof = g.prove_multiple_claims(g.witness(traces, packed)).to_bytes()
    assert proof == o.prove(traces, packed), name
    assert g.verify_multiple_claims(packed, proof) == 0 and o.verify(packed, proof) == 0
    assert g.prove_multiple_claims(g.host_witness(traces, packed)).to_bytes() == proof
    wrong = [list(c) for c in claims]
    wrong[0][-1] ^= 1
    assert g.verify_multiple_claims(fe.pack_claims(wrong), proof) != 0


@pytest.mark.gpu
def test_gpu_whole_hash(pkg, ctx, fe, b3, compiled, oracle, monkeypatch):
    infos, dig = b3.blake3_compressions(bytes(range(256)) * 11 + bytes(184))
    claims = [b3.compression_claim(i) for i in infos]
    traces = b3.blake3_witness(claims)
    packed = fe.pack_claims(claims)
    g = pkg.System.new(ctx, fe.test_params(), compiled[0])
    o = oracle.System(g.blob)
    proof = g.prove_multiple_claims(g.witness(traces, packed)).to_bytes()
    assert proof == o.prove(traces, packed)
    assert g.verify_multiple_claims(packed, proof) == 0
    # bench parameters (proof of work, blowup 4) and FRI rounds of arity 8 on the same system
    for params in (fe.bench_params(), fe.Params(log_blowup=1, max_log_arity=3, num_queries=30)):
        g2 = pkg.System.new(ctx, params, compiled[0])
        p2 = g2.prove_multiple_claims(g2.witness(traces, packed)).to_bytes()
        assert p2 == oracle.System(g2.blob).prove(traces, packed)
        assert g2.verify_multiple_claims(packed, p2) == 0


@pytest.mark.gpu
def test_gpu_joint_proof_over_thread_ranks(pkg, fe, b3, compiled, oracle):
    """the compression circuit (2625 columns) and the G-function circuit owned by different ranks, the tables replicated:
    four thread ranks on the one GPU give the single-GPU proof"""
    sharded = importlib.import_module("multi_stark_amd.sharded")
    infos, _ = b3.blake3_compressions(bytes(range(256)) * 11 + bytes(184))
    claims = [b3.compression_claim(i) for i in infos]
    traces = b3.blake3_witness(claims)
    packed = fe.pack_claims(claims)
    owners = [-1, 1, 2, -1, -1, -1, -1, 3, 0]  # u32 xor / add, G and compression sharded; the byte-pair table and rotations replicated
    world = 4

    def rank_fn(rank, group):
        ctx = pkg.Context(0)
        system = pkg.System.new(ctx, fe.test_params(), compiled[0])
        mine = [t.copy() if owners[i] in (-1, rank) else None for i, t in enumerate(traces)]
        remote = {i: traces[i].shape[0] for i in range(len(traces)) if owners[i] not in (-1, rank)}
        comm = group.comm(ctx, rank)
        try:
            proof = system.prove_sharded(system.witness(mine, packed, remote_heights=remote), comm, owners).to_bytes()
            if rank == 0:
                assert proof == system.prove_multiple_claims(system.witness(traces, packed)).to_bytes()
                assert oracle.System(system.blob).verify(packed, proof) == 0
            return proof
        finally:
            comm.close()

    group = sharded.LocalGroup(world)
    try:
        res = group.run(rank_fn)
    finally:
        group.close()
    assert len(set(res)) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["MSAMD_SIDE_DELAY_US", "MSAMD_MAIN_DELAY_US"])
def test_side_stream_results_are_awaited(pkg, ctx, fe, b3, compiled, oracle, monkeypatch, which):
    """MSAMD_SIDE_DELAY_US: the side stream (short circuits beside long ones) starts 2 ms late at every fork, so a consumer on
    the main stream that does not wait for it reads stale data every time. Found with this system: more than 64 KB of opened
    values are read back by a copy issued at once on the main stream, which did not wait for the side stream's barycentric sums
    (the first proof of a new shape, slowed by fresh allocations, opened the short circuits at wrong values)."""
    monkeypatch.setenv(which, "2000")  # (the second switch delays the main stream behind every fork: the other direction)
    # two witnesses in turn: a stale buffer of the previous proof must not pass for this proof's values
    sets = []
    for data in (bytes(range(256)) * 11 + bytes(184), bytes((7 * i + 1) & 255 for i in range(3000))):
        infos, dig = b3.blake3_compressions(data)
        claims = [b3.compression_claim(i) for i in infos]
        sets.append((b3.blake3_witness(claims), fe.pack_claims(claims)))
    for params in (fe.test_params(), fe.Params(log_blowup=2, num_queries=10), fe.Params(log_blowup=3, cap_height=2, log_final_poly_len=1, num_queries=8,
                                                                                         commit_proof_of_work_bits=2)):
        g = pkg.System.new(ctx, params, compiled[0])
        o = oracle.System(g.blob)
        want = [o.prove(tr, pk) for tr, pk in sets]
        assert want[0] != want[1]
        wits = [g.witness(tr, pk) for tr, pk in sets] + [g.host_witness(tr, pk) for tr, pk in sets]
        for k in (0, 1, 0, 3, 2, 1):
            assert g.prove_multiple_claims(wits[k]).to_bytes() == want[k % 2], (params.log_blowup, k)
    # the bench circuit (the byte table rides the side stream) and the nine-circuit adder system
    for inputs, (tr, cl) in ((fe.u32_add_system_inputs(), fe.u32_add_bench_witness(1 << 14)), (fe.multi_u32_add_system_inputs(8), fe.multi_u32_add_witness(8, 1 << 13))):
        g = pkg.System.new(ctx, fe.bench_params(), inputs)
        pk = fe.pack_claims(cl)
        want = oracle.System(g.blob).prove(tr, pk)
        assert g.prove_multiple_claims(g.witness(tr, pk)).to_bytes() == want
        assert g.prove_multiple_claims(g.host_witness(tr, pk)).to_bytes() == want


@pytest.mark.gpu
@pytest.mark.parametrize("env", [(), ("MSAMD_NO_WAVE_QUOTIENT",), ("MSAMD_NO_WAVE_QUOTIENT", "MSAMD_NO_FEW_LANES"), ("MSAMD_NO_WIDE_PREHASH",),
                                 ("MSAMD_NO_DEEP_WIDE",)])
def test_interpreter_forms_give_the_same_proof(pkg, ctx, fe, b3, compiled, oracle, monkeypatch, env):
    """the compression circuit's 6952-node program is above the hiprtc limit: a short circuit takes the wave-per-row kernel over the
    level-scheduled program (quotient_wave_k); without it few lanes per workgroup with the slot files in LDS; without that the
    thread-per-row interpreter over a global slot file. MSAMD_NO_WIDE_PREHASH: the 2625-column rows injected into the commitment
    trees are hashed by one thread each inside the tree kernel instead of chunk-parallel in front of it (hash.hip)"""
    for e in env:
        monkeypatch.setenv(e, "1")
    infos, dig = b3.blake3_compressions(bytes((5 * i + 2) & 255 for i in range(4000)))
    claims = [b3.compression_claim(i) for i in infos]
    traces = b3.blake3_witness(claims)
    packed = fe.pack_claims(claims)
    for params in (fe.test_params(), fe.Params(log_blowup=2, cap_height=1, num_queries=12, commit_proof_of_work_bits=3)):
        g = pkg.System.new(ctx, params, compiled[0])
        assert g.prove_multiple_claims(g.witness(traces, packed)).to_bytes() == oracle.System(g.blob).prove(traces, packed)
