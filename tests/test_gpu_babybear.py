"""GPU parity of the BabyBear / Poseidon2 path (BASELINE config 4; include/mstark_bb.h) against the oracle compiled for
the same configuration (oracle/libms_oracle_bb.so): kernels at the PCS level, then whole proofs byte for byte - the
reference's own 4-row smoke test (src/test_circuits/baby_bear_config.rs:159-206), parameter variants, lookups with
claims, a preprocessed circuit, mixed heights, and the config-4 size (MulAir at 2^20 rows). The oracle's verifier must
accept every proof. PARITY UNPINNED against Plonky3 itself (see oracle/bb.hpp); the Poseidon2 round constants are inputs."""
import numpy as np
import pytest

import oracle_bb as ob
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
pkg = load_package()
fe = pkg.frontend
bb = pkg.babybear
P = fe.BABYBEAR["P"]
K = fe.poseidon2_constants()


@pytest.fixture(scope="module")
def ctx():
    c = pkg.Context(0)
    bb.set_poseidon2(c, K)
    ob.set_poseidon2(K)
    return c


def rand_field(rng, shape):
    v = rng.integers(0, P, shape, dtype=np.uint64)
    edge = np.array([0, 1, 2, P - 1, P - 2, 1 << 27, (1 << 27) + 1], dtype=np.uint64)
    mask = rng.random(shape) < 0.1
    return np.where(mask, edge[rng.integers(0, len(edge), shape)], v)


def test_field_ops(ctx):
    rng = np.random.default_rng(1)
    a, b = rand_field(rng, 4000), rand_field(rng, 4000)
    A, B = [int(x) for x in a], [int(x) for x in b]
    assert [int(x) for x in bb.field_op(ctx, 0, a, b)] == [(x + y) % P for x, y in zip(A, B)]
    assert [int(x) for x in bb.field_op(ctx, 1, a, b)] == [(x - y) % P for x, y in zip(A, B)]
    assert [int(x) for x in bb.field_op(ctx, 2, a, b)] == [(x * y) % P for x, y in zip(A, B)]
    nz = np.where(a == 0, 1, a)
    assert all(int(x) * int(y) % P == 1 for x, y in zip(nz, bb.field_op(ctx, 3, nz)))
    L = ob.lib()
    qa, qb = rand_field(rng, (500, 4)), rand_field(rng, (500, 4))
    prod = bb.field_op(ctx, 4, qa, qb).reshape(-1, 4)
    inv = bb.field_op(ctx, 5, qa).reshape(-1, 4)
    for i in range(500):
        o = np.zeros(4, dtype=np.uint64)
        L.mso_e2_mul(ob._p(qa[i].copy()), ob._p(qb[i].copy()), ob._p(o))
        assert [int(x) for x in prod[i]] == [int(x) for x in o]
        L.mso_e2_inv(ob._p(qa[i].copy()), ob._p(o))
        assert [int(x) for x in inv[i]] == [int(x) for x in o]


def test_poseidon2_permutation(ctx):
    rng = np.random.default_rng(2)
    st = rand_field(rng, (300, 16))
    st[0] = 0
    st[1] = P - 1
    got = bb.poseidon2_permute(ctx, st)
    for i in range(300):
        assert [int(x) for x in got[i]] == [int(x) for x in ob.poseidon2_permute(st[i])]


@pytest.mark.parametrize("log_h", [0, 1, 3, 7, 11, 12, 13, 15, 16, 19, 20, 21, 22])
def test_dft_batch(ctx, log_h):
    rng = np.random.default_rng(log_h)
    w = 3 if log_h > 14 else 9
    m = rand_field(rng, (1 << log_h, w))
    for inv in (False, True):
        assert np.array_equal(bb.dft_batch(ctx, m, inverse=inv), ob.dft_batch(m, inverse=inv)), (log_h, inv)


@pytest.mark.parametrize("log_h", [13, 16, 21])
def test_dft_batch_lds_passes(ctx, monkeypatch, log_h):
    """the plain LDS radix-2 passes (MSBB_NTT_LDS=1) that the register passes replaced"""
    monkeypatch.setenv("MSBB_NTT_LDS", "1")
    m = rand_field(np.random.default_rng(log_h), (1 << log_h, 2))
    assert np.array_equal(bb.dft_batch(ctx, m), ob.dft_batch(m))


@pytest.mark.parametrize("log_h,lb", [(0, 1), (2, 1), (5, 3), (10, 2), (12, 1), (13, 2), (16, 1), (18, 2)])
def test_coset_lde(ctx, log_h, lb):
    rng = np.random.default_rng(100 + log_h)
    m = rand_field(rng, (1 << log_h, 2 if log_h > 14 else 7))
    assert np.array_equal(bb.coset_lde_batch(ctx, m, lb), ob.coset_lde_bitrev(m, lb))


def _digest_words(b):
    return np.frombuffer(b, dtype=np.uint32)


@pytest.mark.parametrize("shapes,cap_h", [([(8, 3)], 0), ([(1, 5)], 0), ([(64, 8), (64, 9), (16, 1)], 1), ([(256, 24), (128, 3), (2, 17)], 2),
                                          ([(4096, 2), (4096, 40)], 0), ([(1 << 15, 3)], 3)])
def test_mmcs_commit_open(ctx, shapes, cap_h):
    rng = np.random.default_rng(len(shapes) * 7 + cap_h)
    mats = [rand_field(rng, s) for s in shapes]
    g, o = bb.Mmcs(ctx, mats, cap_h), ob.Mmcs(mats, cap_h)
    assert np.array_equal(g.cap, _digest_words(o.cap))
    maxh = max(s[0] for s in shapes)
    for index in {0, maxh - 1, int(rng.integers(0, maxh))}:
        gv, gp = g.open(index)
        ov, op = o.open(index)
        assert np.array_equal(gv, ov) and np.array_equal(gp, _digest_words(op)), index


def _prove_both(ctx, params, inputs, traces, claims):
    with fe.field(fe.BABYBEAR):
        g = bb.System.new(ctx, params, inputs, K)
        o = ob.System(g.blob)
        packed = fe.pack_claims(claims)
    got = g.prove_multiple_claims(g.witness(traces, packed)).to_bytes()
    want = o.prove(traces, packed)
    assert len(got) == len(want)
    assert got == want, "proof bytes differ at byte %d" % next(i for i in range(len(got)) if got[i] != want[i])
    # the same from a witness that stays in host memory (msbb_witness_create_host: upload inside every proof), twice
    hw = g.host_witness(traces, packed)
    assert g.prove_multiple_claims(hw).to_bytes() == want and g.prove_multiple_claims(hw).to_bytes() == want, "host-resident witness: proof differs"
    verdict = o.verify(packed, got)
    assert g.verify(packed, got) == verdict, "product verifier (msbb_verify) disagrees with the oracle's"
    return verdict, g, o, packed, got


def test_reference_smoke_test(ctx):
    """baby_bear_config.rs:159-206"""
    with fe.field(fe.BABYBEAR):
        inputs, params, trace = fe.mul_air_inputs(), fe.test_params(), fe.mul_air_smoke_trace()
    verdict, g, o, packed, proof = _prove_both(ctx, params, inputs, [trace], [])
    assert verdict == 0
    assert g.circuit_info(0) == o.circuit_info(0)
    bad = bytearray(proof)
    bad[len(bad) // 2] ^= 4
    assert o.verify(packed, bytes(bad)) != 0
    # the reference's own tamper: intermediate_accumulators[0] += ONE (baby_bear_config.rs:199-203), field-wise
    import proof_codec as pc

    t = pc.parse(proof, 4, 4)
    t["intermediate_accumulators"][0][0] = (t["intermediate_accumulators"][0][0] + (1 << 32) % P) % P
    tb = pc.serialize(t, 4, 4)
    assert g.verify(packed, tb) == o.verify(packed, tb) == 6


@pytest.mark.parametrize("params", [fe.Params(1, 0, 0, 1, 30, 0, 0), fe.Params(2, 1, 2, 1, 12, 3, 5), fe.Params(3, 2, 1, 1, 8, 6, 0),
                                    fe.Params(2, 0, 0, 1, 100, 10, 10)])
@pytest.mark.parametrize("log_rows", [2, 9, 13])
def test_mul_air_parameter_variants(ctx, params, log_rows):
    if params.log_final_poly_len and log_rows <= params.log_final_poly_len:
        pytest.skip("trace shorter than the final polynomial")
    with fe.field(fe.BABYBEAR):
        inputs, trace = fe.mul_air_inputs(), fe.mul_air_trace(1 << log_rows)
    assert _prove_both(ctx, params, inputs, [trace], [])[0] == 0


def test_lookups_with_claims_and_dead_circuit(ctx):
    """src/lookup.rs:868-1007 (even/odd system + claim [0, 4, 1]) authored over BabyBear"""
    with fe.field(fe.BABYBEAR):
        inputs, traces = fe.even_odd_inputs(), fe.even_odd_traces()
    assert _prove_both(ctx, fe.test_params(), inputs, traces, [[0, 4, 1]])[0] == 0
    assert _prove_both(ctx, fe.test_params(), inputs, traces, [[0, 4, 0]])[0] != 0  # same bytes, both reject


def test_preprocessed_and_mixed_heights(ctx):
    """[ByteTable (preprocessed, 256 rows), U32Add (2^12 rows)] of benches/multi_stark.rs:73-165 over BabyBear, and the
    squares system (constraint degree 3 -> quotient degree 2)"""
    with fe.field(fe.BABYBEAR):
        inputs = fe.u32_add_system_inputs()
        traces, claims = fe.u32_add_bench_witness(1 << 12)  # 4096 claims: accumulated on the device
        assert _prove_both(ctx, fe.Params(2, 0, 0, 1, 20, 2, 2), inputs, traces, claims)[0] == 0
        assert _prove_both(ctx, fe.Params(2, 1, 1, 1, 10, 0, 0), fe.squares_inputs(), fe.squares_traces(16), [])[0] == 0


def test_product_verifier_agrees_with_oracle_on_corrupted_proofs(ctx):
    """msbb_verify vs the oracle's verifier: bit flips, truncation, extension, swapped claims - same accept / reject"""
    with fe.field(fe.BABYBEAR):
        inputs, traces = fe.even_odd_inputs(), fe.even_odd_traces()
        g = bb.System.new(ctx, fe.Params(2, 1, 1, 1, 9, 3, 4), inputs, K)
        o = ob.System(g.blob)
        packed = fe.pack_claims([[0, 4, 1]])
    proof = g.prove_multiple_claims(g.witness(traces, packed)).to_bytes()
    assert g.verify(packed, proof) == 0 and o.verify(packed, proof) == 0
    rng = np.random.default_rng(8)
    for k in range(300):
        bad = bytearray(proof)
        kind = k % 4
        if kind == 0:
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            pos = int(rng.integers(0, len(bad) - 8))
            bad[pos:pos + 8] = int(rng.choice([0, 1, 0xFFFFFFFF, 1 << 40])).to_bytes(8, "little")
        elif kind == 2:
            bad = bad[: int(rng.integers(0, len(bad)))]
        else:
            bad += bytes(int(x) for x in rng.integers(0, 256, 5))
        a, b = g.verify(packed, bytes(bad)), o.verify(packed, bytes(bad))
        assert a != 0 and (a == 0) == (b == 0), (k, a, b)
    with fe.field(fe.BABYBEAR):
        for claims in ([[0, 4, 0]], [], [[0, 4, 1], [0, 4, 1]]):
            pc = fe.pack_claims(claims)
            assert g.verify(pc, proof) != 0 and o.verify(pc, proof) != 0


@pytest.mark.parametrize("var", ["", "MSBB_HOST_FRI", "MSBB_NO_FRI_FUSED", "MSBB_NO_SUBTREE"])
def test_device_side_fri_transcript(ctx, monkeypatch, var):
    """The commit-phase challenger steps run on the device (duplex sponge on 16 lanes; replayed and checked by the host) and a
    round is one launch (fold + leaf digests + tree + challenger step); MSBB_HOST_FRI / MSBB_NO_FRI_FUSED / MSBB_NO_SUBTREE
    select the host-driven rounds, the unfused rounds and the layer-by-layer trees: same bytes every way. With
    proof-of-work bits the host-driven rounds are used regardless."""
    if var:
        monkeypatch.setenv(var, "1")
    with fe.field(fe.BABYBEAR):
        inputs, trace = fe.mul_air_inputs(), fe.mul_air_trace(1 << 11)
        assert _prove_both(ctx, fe.test_params(), inputs, [trace], [])[0] == 0
        assert _prove_both(ctx, fe.Params(2, 2, 1, 1, 9, 0, 3), inputs, [trace], [])[0] == 0   # caps of 4 digests, final poly of 2
        assert _prove_both(ctx, fe.Params(2, 0, 0, 1, 9, 4, 0), inputs, [trace], [])[0] == 0   # PoW: host rounds
        ev_in, ev_tr = fe.even_odd_inputs(), fe.even_odd_traces()
        big = fe.mul_air_trace(1 << 17)   # rounds above the fused kernel's size, trees with 1024 sub-tree roots
        assert _prove_both(ctx, fe.test_params(), inputs, [big], [])[0] == 0
    assert _prove_both(ctx, fe.test_params(), ev_in, ev_tr, [[0, 4, 1]])[0] == 0


def test_quotient_row_chunks(ctx, monkeypatch):
    """the interpreter quotient kernel (what runs without hiprtc, above 3000 nodes and under MSAMD_NO_JIT) with its sweep
    in several row chunks (what a very large trace x node-program product triggers)"""
    monkeypatch.setenv("MSAMD_NO_JIT", "1")
    monkeypatch.setenv("MSBB_QUOTIENT_CHUNK", "192")
    with fe.field(fe.BABYBEAR):
        assert _prove_both(ctx, fe.Params(2, 0, 0, 1, 10, 0, 0), fe.squares_inputs(), fe.squares_traces(256), [])[0] == 0  # 1024 rows
        inputs, trace = fe.mul_air_inputs(), fe.mul_air_trace(1 << 9)
    assert _prove_both(ctx, fe.test_params(), inputs, [trace], [])[0] == 0


def test_interpreter_quotient_with_lookups(ctx, monkeypatch):
    """MSAMD_NO_JIT=1: the per-circuit hiprtc quotient kernels (the default) replaced by the interpreter, on systems with
    lookups, claims, a preprocessed table and mixed heights - same bytes as the oracle either way"""
    monkeypatch.setenv("MSAMD_NO_JIT", "1")
    with fe.field(fe.BABYBEAR):
        inputs, traces = fe.even_odd_inputs(), fe.even_odd_traces()
        assert _prove_both(ctx, fe.test_params(), inputs, traces, [[0, 4, 1]])[0] == 0
        inputs = fe.u32_add_system_inputs()
        traces, claims = fe.u32_add_bench_witness(1 << 8)
        assert _prove_both(ctx, fe.Params(2, 0, 0, 1, 20, 2, 2), inputs, traces, claims)[0] == 0


def test_error_paths(ctx):
    with fe.field(fe.BABYBEAR):
        g = bb.System.new(ctx, fe.test_params(), fe.mul_air_inputs(), K)
        packed = fe.pack_claims([])
        with pytest.raises(pkg.MstarkError):
            g.witness([np.zeros((3, 3), dtype=np.uint64)], packed)  # height not a power of two
        with pytest.raises(pkg.MstarkError):
            g.witness([np.full((4, 3), P, dtype=np.uint64)], packed)  # non-canonical
        with pytest.raises(pkg.MstarkError):
            g.prove_multiple_claims(g.witness([np.zeros((0, 3), dtype=np.uint64)], packed))  # every circuit inactive
        blob = bytearray(g.blob)
        blob[0] ^= 1
        with pytest.raises(pkg.MstarkError):
            bb.System(ctx, bytes(blob), 1)
    with pytest.raises(pkg.MstarkError):  # a Goldilocks blob is refused by the BabyBear entry point
        bb.System(ctx, pkg.System.new(ctx, fe.test_params(), fe.pythagorean_inputs()).blob, 1)


def test_config4_full_size(ctx):
    """BASELINE config 4: MulAir at 2^20 rows, the test-suite parameters of the reference (blowup 2, 64 queries)"""
    with fe.field(fe.BABYBEAR):
        inputs, trace = fe.mul_air_inputs(), fe.mul_air_trace(1 << 20)
        g = bb.System.new(ctx, fe.test_params(), inputs, K)
        o = ob.System(g.blob)
        packed = fe.pack_claims([])
    w = g.witness([trace], packed)
    proof = g.prove_multiple_claims(w, want_times=True)
    print("config 4 (2^20 rows) stage ms:", {k: round(v, 1) for k, v in proof.stage_ms.items()})
    assert o.verify(packed, proof.to_bytes()) == 0
    assert o.prove([trace], packed) == proof.to_bytes()
    hw = g.host_witness([trace], packed)
    for _ in range(3):
        assert g.prove_multiple_claims(hw).to_bytes() == proof.to_bytes()
    assert g.prove_multiple_claims(w).to_bytes() == proof.to_bytes()  # and the device-resident one again on the same context


@pytest.mark.parametrize("kw", [dict(log_blowup=1, max_log_arity=2), dict(log_blowup=2, max_log_arity=3, commit_proof_of_work_bits=3, query_proof_of_work_bits=2),
                                dict(log_blowup=1, max_log_arity=4, log_final_poly_len=1, cap_height=1), dict(log_blowup=1, max_log_arity=6)])
def test_wide_fri_folds(ctx, kw):
    """FRI rounds of arity up to 2^max_log_arity (src/types.rs:189-190; baby_bear_config.rs:63,79 passes the field through):
    mixed heights (the byte table's roll-in bounds a round's arity), lookups + claims, the MulAir; bytes == oracle, both
    verifiers accept and agree on corrupted proofs; a binary-fold system refuses the bytes"""
    params = fe.Params(num_queries=20, **kw)
    with fe.field(fe.BABYBEAR):
        inputs = fe.u32_add_system_inputs()
        traces, claims = fe.u32_add_bench_witness(1 << 9)
        verdict, g, o, packed, proof = _prove_both(ctx, params, inputs, traces, claims)
        assert verdict == 0
        rng = np.random.default_rng(17)
        rejected = 0
        for pos in [int(x) for x in rng.integers(0, len(proof), 40)]:
            bad = bytearray(proof)
            bad[pos] ^= 1 << int(rng.integers(0, 8))
            v = o.verify(packed, bytes(bad))
            assert (v != 0) == (g.verify(packed, bytes(bad)) != 0), pos
            rejected += v != 0
        assert rejected >= 36  # (a proof-of-work witness is not read at zero bits: a flip there is accepted by both)
        other = bb.System.new(ctx, fe.Params(num_queries=20, **dict(kw, max_log_arity=1)), inputs, K)
        assert other.verify(packed, proof) != 0
        assert _prove_both(ctx, params, fe.even_odd_inputs(), fe.even_odd_traces(), [[0, 4, 1]])[0] == 0
        assert _prove_both(ctx, params, fe.mul_air_inputs(), [fe.mul_air_trace(1 << 11)], [])[0] == 0
        with pytest.raises(pkg.MstarkError):
            bb.System.new(ctx, fe.Params(max_log_arity=7), fe.mul_air_inputs(), K)
