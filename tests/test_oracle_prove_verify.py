"""CPU tests: the oracle's prover/verifier on the reference's end-to-end scenarios (prove -> verify, tamper
rejection, serialisation round trip) and against the committed golden fixtures."""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "oracle_refs.json")))


def _system(oracle, fe, inputs, params):
    comp = [fe.compile_circuit(ci) for ci in inputs]
    blob = fe.system_blob(params, comp)
    return oracle.System(blob), blob, comp


def _check_gold(name, blob, proof, n_circuits):
    g = GOLD["proofs"][name]
    assert hashlib.sha256(blob).hexdigest() == g["blob_sha256"]
    assert len(proof) == g["proof_len"]
    assert hashlib.sha256(proof).hexdigest() == g["proof_sha256"]
    assert proof[8 + n_circuits + 8: 8 + n_circuits + 40].hex() == g["stage1_commit"]


def test_golden_pcs_and_challenger_refs(oracle):
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    assert mg.pcs_refs() == GOLD["pcs_refs"]
    assert mg.challenger_refs() == GOLD["challenger_refs"]


# examples/simple_proof.rs:46-91 (config 1) — 4 rows as in the example and the north-star 2^12 rows
@pytest.mark.parametrize("rows,name", [(4, "simple_proof_4"), (4096, "simple_proof_4096")])
def test_simple_proof(oracle, fe, rows, name):
    s, blob, comp = _system(oracle, fe, fe.pythagorean_inputs(), fe.test_params())
    packed = fe.pack_claims([])
    proof = s.prove([fe.pythagorean_trace(rows)], packed)
    assert s.verify(packed, proof) == 0
    _check_gold(name, blob, proof, 1)
    # determinism (src/types.rs:31-42: zero-bit PoW must not make proofs run-dependent)
    assert s.prove([fe.pythagorean_trace(rows)], packed) == proof


def test_unsatisfied_constraint_is_rejected(oracle, fe):
    s, _, _ = _system(oracle, fe, fe.pythagorean_inputs(), fe.test_params())
    tr = fe.pythagorean_trace(8)
    tr[3, 2] += 1  # 8^2 + 15^2 != 18^2
    packed = fe.pack_claims([])
    assert s.verify(packed, s.prove([tr], packed)) != 0


# src/verifier.rs:852-912 style tampering: any flipped byte of the proof must be rejected
def test_tampered_proofs_rejected(oracle, fe):
    s, _, _ = _system(oracle, fe, fe.pythagorean_inputs(), fe.test_params())
    packed = fe.pack_claims([])
    proof = s.prove([fe.pythagorean_trace(16)], packed)
    rng = np.random.default_rng(5)
    for pos in [9, 20, 60] + [int(x) for x in rng.integers(0, len(proof), 40)]:
        bad = bytearray(proof)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        assert s.verify(packed, bytes(bad)) != 0, "tampering at byte %d accepted" % pos
    assert s.verify(packed, proof[:-1]) != 0 and s.verify(packed, proof + b"\0") != 0


# src/lookup.rs:1043-1051 and :1079-1130
def test_lookup_proof_and_wrong_claims(oracle, fe):
    s, blob, comp = _system(oracle, fe, fe.even_odd_inputs(), fe.test_params())
    packed = fe.pack_claims([[0, 4, 1]])
    proof = s.prove(fe.even_odd_traces(), packed)
    assert s.verify(packed, proof) == 0
    _check_gold("lookup_even_odd", blob, proof, 2)
    for bad in ([[0, 4, 0]], [[0, 4], [1]], [], [[0, 4, 1], [0, 4, 1]]):
        assert s.verify(fe.pack_claims(bad), proof) != 0
    # a proof for an unbalanced claim is produced but rejected (UnbalancedChannel = 6)
    wrong = fe.pack_claims([[0, 4, 0]])
    assert s.verify(wrong, s.prove(fe.even_odd_traces(), wrong)) == 6


def test_sparse_inactive_circuit(oracle, fe):
    s, _, _ = _system(oracle, fe, fe.even_odd_inputs(with_dead=True), fe.test_params())
    packed = fe.pack_claims([[0, 4, 1]])
    traces = fe.even_odd_traces() + [np.zeros((0, 6), dtype=np.uint64)]
    proof = s.prove(traces, packed)
    assert s.verify(packed, proof) == 0
    assert proof[8:11] == b"\x01\x01\x00"  # activation bitmap
    # flipping the bitmap must be rejected (src/lookup.rs: bitmap tamper)
    bad = bytearray(proof)
    bad[10] = 1
    assert s.verify(packed, bytes(bad)) != 0


# src/test_circuits/u32_add.rs:193-221
def test_u32_add_proof(oracle, fe):
    s, blob, comp = _system(oracle, fe, fe.u32_add_system_inputs(), fe.test_params())
    traces, claims = fe.u32_add_witness([(10, 5), (30, 20), (100, 100), (8000, 10000)])
    packed = fe.pack_claims(claims)
    proof = s.prove(traces, packed)
    assert s.verify(packed, proof) == 0
    _check_gold("u32_add_proof", blob, proof, 2)
    bad = claims.copy()
    bad[2, 3] = 201
    assert s.verify(fe.pack_claims(bad), proof) != 0


# src/test_circuits/byte_operations.rs:124-157 (byte_test): 2^16-row preprocessed table of width 5, no AIR constraints,
# lookups with 4 and 3 arguments, claims of different lengths
def test_byte_operations_proof(oracle, fe):
    s, _, _ = _system(oracle, fe, fe.byte_operations_inputs(), fe.test_params())
    calls = [(0, 10, 5), (1, 30, 20), (2, 100, 40), (3, 200, 100)]
    traces, claims = fe.byte_operations_witness(calls)
    assert claims == [[0, 10, 5, 15], [1, 30, 20, 20], [2, 100, 40, 108], [3, 200, 100]]
    packed = fe.pack_claims(claims)
    proof = s.prove(traces, packed)
    assert s.verify(packed, proof) == 0
    wrong = fe.pack_claims([[0, 10, 5, 14]] + claims[1:])
    assert s.verify(wrong, proof) != 0


# examples/preprocessed_proof.rs: preprocessed range table + squaring circuit, one-argument lookups, no claims
@pytest.mark.parametrize("n", [16, 256])
def test_preprocessed_squares_proof(oracle, fe, n):
    s, _, _ = _system(oracle, fe, fe.squares_inputs(), fe.test_params())
    traces = fe.squares_traces(n)
    packed = fe.pack_claims([])
    proof = s.prove(traces, packed)
    assert s.verify(packed, proof) == 0
    traces[1][3, 1] += 1  # x^2 off by one: the AIR constraint fails, the verifier rejects
    assert s.verify(packed, s.prove(traces, packed)) != 0


def test_bench_workload_small(oracle, fe):
    s, blob, comp = _system(oracle, fe, fe.u32_add_system_inputs(), fe.bench_params())
    traces, claims = fe.u32_add_bench_witness(1 << 8)
    packed = fe.pack_claims(claims)
    proof, times = s.prove(traces, packed, want_times=True)
    assert s.verify(packed, proof) == 0
    _check_gold("bench_u32_add_2p8", blob, proof, 2)
    assert times["total"] > 0


def test_degree_limit_and_bad_blobs(oracle, fe):
    # src/system.rs:404-445: x^5 = y needs quotient degree 4 > 2^log_blowup at log_blowup 1, fine at 2
    def ev(b):
        local, _ = b.main()
        x = local[0]
        b.assert_eq(x * x * x * x * x, local[1])

    inputs = [fe.lookup_air(2, ev, [])]
    comp = [fe.compile_circuit(ci) for ci in inputs]
    with pytest.raises(RuntimeError):
        oracle.System(fe.system_blob(fe.Params(log_blowup=1), comp))
    s = oracle.System(fe.system_blob(fe.Params(log_blowup=2, num_queries=30), comp))
    assert s.circuit_info(0)["quotient_degree"] == 4
    tr = np.array([[i, pow(i, 5, fe.P)] for i in range(8)], dtype=np.uint64)
    packed = fe.pack_claims([])
    assert s.verify(packed, s.prove([tr], packed)) == 0
    with pytest.raises(RuntimeError):
        oracle.System(b"\0" * 64)
    # height mismatch with a preprocessed trace (src/system.rs:447-467)
    s2, _, _ = _system(oracle, fe, fe.u32_add_system_inputs(), fe.test_params())
    with pytest.raises(RuntimeError):
        s2.prove([np.zeros((128, 1), dtype=np.uint64), np.zeros((4, 14), dtype=np.uint64)], fe.pack_claims([]))
