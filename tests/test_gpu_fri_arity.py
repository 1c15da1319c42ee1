"""GPU tests: FRI rounds of arity above 2 (`FriParameters::max_log_arity`, /root/reference/src/types.rs:189-190,215) in the HIP
prover and the product's verifier, through the C ABI, bit-exact against the oracle (whose fold is itself checked against Lagrange
interpolation in tests/test_oracle_fri_arity.py). The product's verifier folds a row by the barycentric formula over the row's
coset, the oracle's by binary steps, the prover by repeated binary fold launches: three forms that have to agree."""
import numpy as np
import pytest

from test_gpu_prove import _prove_both
from test_gpu_kernels import _pcs_scenario
from conftest import rand_field

pytestmark = pytest.mark.gpu

WIDE = [dict(log_blowup=1, max_log_arity=2), dict(log_blowup=2, max_log_arity=3, commit_proof_of_work_bits=3, query_proof_of_work_bits=2),
        dict(log_blowup=1, max_log_arity=4, log_final_poly_len=1), dict(log_blowup=2, cap_height=2, max_log_arity=2),
        dict(log_blowup=1, max_log_arity=6), dict(log_blowup=3, max_log_arity=5, log_final_poly_len=1, cap_height=1)]


@pytest.mark.parametrize("kw", WIDE)
def test_whole_proofs_with_wide_folds(pkg, ctx, oracle, fe, kw):
    params = fe.Params(num_queries=25, **kw)
    # mixed heights: the shorter trace's roll-in bounds a round's arity (even/odd lookups, src/lookup.rs:1043-1051)
    g, o, packed, proof = _prove_both(pkg, ctx, oracle, fe, fe.even_odd_inputs(), params, fe.even_odd_traces(), [[0, 4, 1]])
    rng = np.random.default_rng(3)
    rejected = 0
    for pos in [int(x) for x in rng.integers(0, len(proof), 40)]:
        bad = bytearray(proof)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        v = o.verify(packed, bytes(bad))
        assert (g.verify_multiple_claims(packed, bytes(bad)) != 0) == (v != 0), "verifiers disagree on tampering at byte %d" % pos
        rejected += v != 0
    assert rejected >= 36  # (a proof-of-work witness is not read at zero bits: a flip there is accepted by both)
    # the bench circuit: several rounds at the full arity, the byte table rolled in on the way down
    traces, claims = fe.u32_add_bench_witness(1 << 9)
    g, o, packed, proof = _prove_both(pkg, ctx, oracle, fe, fe.u32_add_system_inputs(), params, traces, claims)
    # host-resident witness and a second proof from the same handles
    assert g.prove_multiple_claims(g.host_witness(traces, packed)).to_bytes() == proof
    # a system configured for binary folds refuses the bytes
    other = pkg.System.new(ctx, fe.Params(num_queries=25, **dict(kw, max_log_arity=1)), fe.u32_add_system_inputs())
    assert other.verify_multiple_claims(packed, proof) != 0
    # preprocessed trace (examples/preprocessed_proof.rs)
    _prove_both(pkg, ctx, oracle, fe, fe.squares_inputs(), params, fe.squares_traces(64), [])


def test_bench_parameters_with_arity_8_at_2_pow_14(pkg, ctx, oracle, fe):
    """bench_config() (benches/multi_stark.rs:244-258) with max_log_arity = 3: 2^16-row vectors, proof of work per round"""
    params = fe.Params(2, 0, 0, 3, 100, 10, 10)
    traces, claims = fe.u32_add_bench_witness(1 << 14)
    g, o, packed, proof = _prove_both(pkg, ctx, oracle, fe, fe.u32_add_system_inputs(), params, traces, claims)
    binary = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    assert len(proof) < len(binary.prove_multiple_claims(binary.witness(traces, packed)).to_bytes())  # fewer, wider openings


@pytest.mark.parametrize("params_kw", [dict(log_blowup=2, cap_height=1, log_final_poly_len=1, max_log_arity=2, num_queries=15, commit_proof_of_work_bits=4,
                                            query_proof_of_work_bits=5),
                                       dict(log_blowup=1, max_log_arity=3, num_queries=30), dict(log_blowup=1, max_log_arity=6, num_queries=10)])
def test_pcs_commit_open_verify_with_wide_folds(ctx, pkg, oracle, fe, params_kw):
    """Pcs::commit / open / verify on their own (examples/pcs_example.rs) - opened values, FriProof bytes, both verifiers"""
    params = fe.Params(**params_kw)
    rng = np.random.default_rng(params.num_queries)
    rounds = [[rand_field(rng, (1 << 10, 5)), rand_field(rng, (1 << 7, 2)), rand_field(rng, (1 << 10, 1))], [rand_field(rng, (1 << 9, 9))]]
    _pcs_scenario(ctx, pkg, oracle, params, rounds, -1)
    _pcs_scenario(ctx, pkg, oracle, params, [[np.arange(32, dtype=np.uint64).reshape(32, 1)]], 2)


def test_arity_above_64_is_refused(pkg, ctx, fe):
    with pytest.raises(pkg.MstarkError):
        pkg.System.new(ctx, fe.Params(max_log_arity=7), fe.pythagorean_inputs())
    with pytest.raises(pkg.MstarkError):
        pkg.System.new(ctx, fe.Params(max_log_arity=0), fe.pythagorean_inputs())


def test_random_systems_with_wide_folds(pkg, ctx, oracle, fe, monkeypatch):
    """a seeded slice of tools/fuzz_parity.py with FUZZ_ARITY=1: random systems and parameters, max_log_arity drawn from 1..6"""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity

    monkeypatch.setenv("MSAMD_NO_JIT", "1")
    monkeypatch.setenv("FUZZ_ARITY", "1")
    rng = np.random.default_rng(777)
    tally = {}
    for case in range(80):
        r = fuzz_parity.one_case(pkg, fe, oracle, ctx, np.random.default_rng(rng.integers(0, 1 << 62)), case)
        tally[r] = tally.get(r, 0) + 1
    assert tally.get("proved", 0) + tally.get("verified", 0) >= 40, tally


def test_wide_rounds_do_not_wait_for_the_host(pkg, ctx, oracle, fe, monkeypatch):
    """rounds of arity above 2 take the device transcript (round 4): a proof at the bench parameters with max_log_arity = 3 waits
    for the host as often as a binary one (twice); host-driven (MSAMD_HOST_WIDE_FRI=1) it waits once more per round. Same bytes."""
    params = fe.Params(2, 0, 0, 3, 100, 10, 10)
    traces, claims = fe.u32_add_bench_witness(1 << 12)
    g, o, packed, proof = _prove_both(pkg, ctx, oracle, fe, fe.u32_add_system_inputs(), params, traces, claims)
    w = g.witness(traces, packed)

    def waits():
        n0 = ctx.sync_count()
        assert g.prove_multiple_claims(w).to_bytes() == proof
        return ctx.sync_count() - n0

    on_device = waits()
    monkeypatch.setenv("MSAMD_HOST_WIDE_FRI", "1")
    host_driven = waits()
    assert on_device == 2 and host_driven >= on_device + 4, (on_device, host_driven)
