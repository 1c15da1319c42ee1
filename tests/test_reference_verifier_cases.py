"""The reference verifier's own test cases, one for one (src/verifier.rs:784-920, src/lookup.rs:1043-1131): the system
[Pythagorean (degree 3), Complex], its serialisation round trip, and the FIELD-wise tampering of the negative tests
(tests/proof_codec.py parses and re-serialises Proof::to_bytes). CPU tests run them on the oracle; the `gpu` tests run
them through the C ABI: ms_prove must emit the oracle's bytes and ms_verify must give the oracle's verdicts."""
import numpy as np
import pytest

import proof_codec as pc

P = 0xFFFFFFFF00000001


def _small(fe, doublings=0):
    return fe.verifier_test_inputs(), fe.test_params(), fe.verifier_test_traces(doublings)


def _tamper_cases(proof_bytes):
    """name -> (tampered bytes, claims) for src/verifier.rs:852-912"""
    out = {}
    p = pc.parse(proof_bytes)
    assert pc.serialize(p) == proof_bytes  # test_serialization_round_trip: bytes -> structure -> same bytes
    out["wrong_claim"] = (proof_bytes, [[42]])
    t = pc.parse(proof_bytes)
    t["stage_1_opened_values"][0][0][0][0] = (t["stage_1_opened_values"][0][0][0][0] + 1) % P
    out["tampered_stage_1_values"] = (pc.serialize(t), [])
    t = pc.parse(proof_bytes)
    t["intermediate_accumulators"][-1] = [1, 0]
    out["tampered_accumulator"] = (pc.serialize(t), [])
    t = pc.parse(proof_bytes)
    t["log_degrees"].pop()
    out["truncated_log_degrees"] = (pc.serialize(t), [])
    t = pc.parse(proof_bytes)
    t["log_degrees"][0] = 200
    out["oversized_log_degree"] = (pc.serialize(t), [])
    t = pc.parse(proof_bytes)
    t["quotient_opened_values"].pop()
    out["truncated_proof"] = (pc.serialize(t), [])
    return out


EXPECT = {"wrong_claim": None, "tampered_stage_1_values": 2, "tampered_accumulator": 6, "truncated_log_degrees": 3, "oversized_log_degree": 3,
          "truncated_proof": 3}  # VerificationError variants (None: any error)


def _check_verdicts(verify, fe, proof):
    for name, (bad, claims) in _tamper_cases(proof).items():
        v = verify(fe.pack_claims(claims), bad)
        assert v != 0, name
        if EXPECT[name] is not None:
            assert v == EXPECT[name], (name, v)


@pytest.mark.parametrize("doublings", [0, 4])
def test_multi_stark_prove_verify_serialize_oracle(oracle, fe, doublings):
    """multi_stark_test (:784) and multi_stark_prove_verify_serialize (:803)"""
    inputs, params, traces = _small(fe, doublings)
    o = oracle.System(fe.system_blob(params, [fe.compile_circuit(c) for c in inputs]))
    assert o.circuit_info(0)["quotient_degree"] == 2 and o.circuit_info(1)["quotient_degree"] == 1
    packed = fe.pack_claims([])
    proof = o.prove(traces, packed)
    assert o.verify(packed, proof) == 0
    assert pc.serialize(pc.parse(proof)) == proof


def test_negative_cases_oracle(oracle, fe):
    inputs, params, traces = _small(fe)
    o = oracle.System(fe.system_blob(params, [fe.compile_circuit(c) for c in inputs]))
    proof = o.prove(traces, fe.pack_claims([]))
    _check_verdicts(o.verify, fe, proof)


def test_sparse_needed_circuit_rejected_oracle(oracle, fe):
    """src/lookup.rs:1100-1115: emptying the Odd table leaves the logUp accumulator unbalanced"""
    o = oracle.System(fe.system_blob(fe.test_params(), [fe.compile_circuit(c) for c in fe.even_odd_inputs()]))
    traces = fe.even_odd_traces()
    traces[1] = np.zeros((0, 6), dtype=np.uint64)
    packed = fe.pack_claims([[0, 4, 1]])
    proof = o.prove(traces, packed)
    assert pc.parse(proof)["active"] == [1, 0]
    assert o.verify(packed, proof) == 6  # UnbalancedChannel


def test_sparse_bitmap_tamper_rejected_oracle(oracle, fe):
    """src/lookup.rs:1077-1098: activating a circuit without data, or deactivating one with data"""
    o = oracle.System(fe.system_blob(fe.test_params(), [fe.compile_circuit(c) for c in fe.even_odd_inputs(with_dead=True)]))
    packed = fe.pack_claims([[0, 4, 1]])
    proof = o.prove(fe.even_odd_traces() + [np.zeros((0, 6), dtype=np.uint64)], packed)
    p = pc.parse(proof)
    assert p["active"] == [1, 1, 0] and len(p["log_degrees"]) == 2 and len(p["intermediate_accumulators"]) == 2
    assert len(p["stage_1_opened_values"]) == 2
    p["active"][2] = 1
    assert o.verify(packed, pc.serialize(p)) != 0
    p["active"][2], p["active"][1] = 0, 0
    assert o.verify(packed, pc.serialize(p)) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("doublings", [0, 4])
def test_multi_stark_prove_verify_serialize_gpu(pkg, ctx, oracle, fe, doublings):
    inputs, params, traces = _small(fe, doublings)
    g = pkg.System.new(ctx, params, inputs)
    o = oracle.System(g.blob)
    packed = fe.pack_claims([])
    proof = g.prove_multiple_claims(g.witness(traces, packed)).to_bytes()
    assert proof == o.prove(traces, packed)
    assert g.verify_multiple_claims(packed, proof) == 0 and o.verify(packed, proof) == 0
    assert pc.serialize(pc.parse(proof)) == proof


@pytest.mark.gpu
def test_negative_cases_gpu(pkg, ctx, oracle, fe):
    inputs, params, traces = _small(fe)
    g = pkg.System.new(ctx, params, inputs)
    proof = g.prove_multiple_claims(g.witness(traces, fe.pack_claims([]))).to_bytes()
    _check_verdicts(g.verify_multiple_claims, fe, proof)
    _check_verdicts(oracle.System(g.blob).verify, fe, proof)


@pytest.mark.gpu
def test_sparse_cases_gpu(pkg, ctx, oracle, fe):
    g = pkg.System.new(ctx, fe.test_params(), fe.even_odd_inputs())
    traces = fe.even_odd_traces()
    traces[1] = np.zeros((0, 6), dtype=np.uint64)
    packed = fe.pack_claims([[0, 4, 1]])
    proof = g.prove_multiple_claims(g.witness(traces, packed)).to_bytes()
    assert proof == oracle.System(g.blob).prove(traces, packed)
    assert pc.parse(proof)["active"] == [1, 0] and g.verify_multiple_claims(packed, proof) == 6
    g3 = pkg.System.new(ctx, fe.test_params(), fe.even_odd_inputs(with_dead=True))
    proof = g3.prove_multiple_claims(g3.witness(fe.even_odd_traces() + [np.zeros((0, 6), dtype=np.uint64)], packed)).to_bytes()
    p = pc.parse(proof)
    assert p["active"] == [1, 1, 0]
    p["active"][2] = 1
    assert g3.verify_multiple_claims(packed, pc.serialize(p)) != 0
    p["active"][2], p["active"][1] = 0, 0
    assert g3.verify_multiple_claims(packed, pc.serialize(p)) != 0
