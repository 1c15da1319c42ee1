"""CPU checks of the oracle compiled for the reference's second configuration (BASELINE config 4:
src/test_circuits/baby_bear_config.rs - BabyBear, degree-4 extension, Poseidon2 sponge / compression, DuplexChallenger).
PARITY UNPINNED against Plonky3 itself (no vectors in the reference; its Poseidon2 constants come from an RNG stream that
cannot be reproduced here and are inputs). Pinned: field constants by arithmetic identities, the permutation against an
independent pure-Python restatement, sponge / compression / challenger semantics, and prove -> verify incl. the
reference's own smoke test (4-row MulAir, tampered proof rejected)."""
import numpy as np
import pytest

import oracle_bb as ob
from __graft_entry__ import load_package

pkg = load_package()
fe = pkg.frontend
P = fe.BABYBEAR["P"]
K = fe.poseidon2_constants()


@pytest.fixture(autouse=True)
def _perm():
    ob.set_poseidon2(K)


def test_field_constants():
    L = ob.lib()
    assert int(L.mso_field_order()) == P == 2013265921
    g27 = int(L.mso_gl_two_adic_generator(27))
    assert g27 == pow(31, (P - 1) >> 27, P) == 0x1A427A41
    assert pow(g27, 1 << 26, P) == P - 1
    for bits in (1, 5, 20):
        assert int(L.mso_gl_two_adic_generator(bits)) == pow(g27, 1 << (27 - bits), P)
    rng = np.random.default_rng(1)
    for _ in range(50):
        a, b = (int(x) for x in rng.integers(0, P, 2))
        assert int(L.mso_gl_mul(a, b)) == a * b % P
        assert int(L.mso_gl_add(a, b)) == (a + b) % P
        assert int(L.mso_gl_sub(a, b)) == (a - b) % P
        if a:
            assert int(L.mso_gl_inv(a)) * a % P == 1
    # serde form of MontyField31: x * 2^32 mod p
    for x in (0, 1, 2, P - 1, 123456789):
        assert int(L.mso_to_wire(x)) == (x << 32) % P


def _ext_mul(a, b):
    out = [0] * 4
    for i in range(4):
        for j in range(4):
            if i + j < 4:
                out[i + j] = (out[i + j] + a[i] * b[j]) % P
            else:
                out[i + j - 4] = (out[i + j - 4] + 11 * a[i] * b[j]) % P
    return out


def test_ext4_mul_and_inverse():
    rng = np.random.default_rng(2)
    L = ob.lib()
    for _ in range(50):
        a = rng.integers(0, P, 4, dtype=np.uint64)
        b = rng.integers(0, P, 4, dtype=np.uint64)
        o = np.zeros(4, dtype=np.uint64)
        L.mso_e2_mul(ob._p(a), ob._p(b), ob._p(o))
        assert [int(x) for x in o] == _ext_mul([int(x) for x in a], [int(x) for x in b])
        L.mso_e2_inv(ob._p(a), ob._p(o))
        assert _ext_mul([int(x) for x in a], [int(x) for x in o]) == [1, 0, 0, 0]
    x = [0, 1, 0, 0]
    x4 = _ext_mul(_ext_mul(x, x), _ext_mul(x, x))
    assert x4 == [11, 0, 0, 0]  # X^4 = W = 11


# ---- independent pure-Python Poseidon2 (width 16, x^7, 4 + 13 + 4 rounds)
M4 = [[2, 3, 1, 1], [1, 2, 3, 1], [1, 1, 2, 3], [3, 1, 1, 2]]
inv = lambda v: pow(v, P - 2, P)  # noqa: E731
V = [-2, 1, 2, inv(2), 3, 4, -inv(2), -3, -4, inv(1 << 8), inv(4), inv(8), inv(1 << 27), -inv(1 << 8), -inv(16), -inv(1 << 27)]


def _mds_light(s):
    t = []
    for c in range(0, 16, 4):
        t += [sum(M4[r][k] * s[c + k] for k in range(4)) % P for r in range(4)]
    col = [sum(t[4 * j + k] for j in range(4)) % P for k in range(4)]
    return [(t[i] + col[i % 4]) % P for i in range(16)]


def _permute(s):
    ext = [[int(x) for x in K[16 * r:16 * r + 16]] for r in range(8)]
    internal = [int(x) for x in K[128:141]]
    s = _mds_light([int(x) for x in s])
    for r in range(4):
        s = _mds_light([pow((s[i] + ext[r][i]) % P, 7, P) for i in range(16)])
    for r in range(13):
        s[0] = pow((s[0] + internal[r]) % P, 7, P)
        tot = sum(s) % P
        s = [(tot + V[i] * s[i]) % P for i in range(16)]
    for r in range(4, 8):
        s = _mds_light([pow((s[i] + ext[r][i]) % P, 7, P) for i in range(16)])
    return s


def test_poseidon2_against_python_restatement():
    rng = np.random.default_rng(3)
    for st in ([0] * 16, list(range(16)), [P - 1] * 16, [int(x) for x in rng.integers(0, P, 16)]):
        assert [int(x) for x in ob.poseidon2_permute(st)] == _permute(st)


def _digest(words):
    return b"".join(int(w).to_bytes(4, "little") for w in words)


def test_sponge_and_compression_semantics():
    assert ob.hash_elems([]) == bytes(32)  # PaddingFreeSponge: no input, no permutation
    for n in (1, 7, 8, 9, 16, 17, 40):
        xs = [(i * i + 3) % P for i in range(n)]
        st = [0] * 16
        for i in range(0, n, 8):
            blk = xs[i:i + 8]
            st[:len(blk)] = blk  # overwrite, not add; a partial block keeps the rest of the state
            st = _permute(st)
        assert ob.hash_elems(xs) == _digest(st[:8]), n
    l, r = list(range(1, 9)), list(range(9, 17))
    assert ob.compress2(_digest(l), _digest(r)) == _digest(_permute(l + r)[:8])  # TruncatedPermutation


def test_duplex_challenger_semantics():
    ch = ob.Challenger(b"")
    for v in (5, 6, 7):
        ch.observe(v)
    st = _permute([5, 6, 7] + [0] * 13)
    # sampling absorbs the 3 queued values, then pops from the back of state[..8]
    assert ch.sample_ext() == (st[7], st[6], st[5], st[4])
    assert ch.sample_bits(10) == st[3] & 1023
    ch.observe(9)  # clears the output buffer; the old state words stay
    st2 = _permute([9] + st[1:])
    assert ch.sample_bits(20) == st2[7] & ((1 << 20) - 1)
    # 8 queued values are absorbed at once
    ch2 = ob.Challenger(b"")
    for v in range(8):
        ch2.observe(v + 1)
    st3 = _permute(list(range(1, 9)) + [0] * 8)
    ch2.observe(77)
    st4 = _permute([77] + st3[1:])
    assert ch2.sample_bits(30) == st4[7] & ((1 << 30) - 1)
    # grinding returns the smallest witness
    ch3 = ob.Challenger(b"abc")
    w = ch3.grind(6)
    for cand in range(w + 1):
        c = ob.Challenger(b"abc")
        c.observe(cand)
        assert (c.sample_bits(6) == 0) == (cand == w)


def test_golden_fixture_babybear():
    """tests/golden/oracle_refs.json["babybear"]: self-generated regression vectors (tests/golden/make_golden.py)"""
    import hashlib
    import json
    import os

    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_refs.json")))["babybear"]
    assert hashlib.sha256(np.ascontiguousarray(K, dtype=np.uint64).tobytes()).hexdigest() == gold["constants_sha256"]
    assert [int(x) for x in ob.poseidon2_permute(np.arange(16))] == gold["permute_0_to_15"]
    assert ob.hash_elems(list(range(1, 12))).hex() == gold["leaf_1_to_11"]
    ch = ob.Challenger(b"multi-stark/v0")
    ch.observe(7)
    assert list(ch.sample_ext()) == gold["challenger_sample_ext"] and ch.sample_bits(20) == gold["challenger_sample_bits_20"]
    with fe.field(fe.BABYBEAR):
        o = _system(fe.test_params(), fe.mul_air_inputs())
        packed = fe.pack_claims([])
        p = o.prove([fe.mul_air_smoke_trace()], packed)
        assert (len(p), hashlib.sha256(p).hexdigest()) == (gold["mul_air_smoke_test"]["proof_len"], gold["mul_air_smoke_test"]["proof_sha256"])
        p = o.prove([fe.mul_air_trace(1 << 10)], packed)
        assert hashlib.sha256(p).hexdigest() == gold["mul_air_1024"]["proof_sha256"]


def _system(params, inputs):
    comp = [fe.compile_circuit(c) for c in inputs]
    return ob.System(fe.system_blob(params, comp, K))


def test_reference_smoke_test_mul_air():
    """baby_bear_config.rs:159-206: 4 rows, blowup 2, 64 queries, no proof of work; a tampered proof is rejected"""
    with fe.field(fe.BABYBEAR):
        o = _system(fe.test_params(), fe.mul_air_inputs())
        info = o.circuit_info(0)
        assert info["stage2_width"] == 8 and info["constraint_count"] == 1 + 2 * 4 and info["quotient_degree"] == 1
        packed = fe.pack_claims([])
        proof = o.prove([fe.mul_air_smoke_trace()], packed)
        assert o.verify(packed, proof) == 0
        assert o.prove([fe.mul_air_smoke_trace()], packed) == proof  # deterministic
        rng = np.random.default_rng(4)
        for _ in range(40):
            bad = bytearray(proof)
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
            assert o.verify(packed, bytes(bad)) != 0
        # the reference's own tamper: tampered.intermediate_accumulators[0] += Challenge::ONE (baby_bear_config.rs:199-203)
        import proof_codec as pc

        t = pc.parse(proof, 4, 4)
        assert pc.serialize(t, 4, 4) == proof
        one_monty = (1 << 32) % P
        t["intermediate_accumulators"][0][0] = (t["intermediate_accumulators"][0][0] + one_monty) % P
        assert o.verify(packed, pc.serialize(t, 4, 4)) == 6  # UnbalancedChannel: the last accumulator is no longer zero
        wrong = fe.mul_air_smoke_trace()
        wrong[1, 2] = 21  # 4 * 5 != 21
        assert o.verify(packed, o.prove([wrong], packed)) == 5  # OodEvaluationMismatch


@pytest.mark.parametrize("params", [fe.Params(1, 0, 0, 1, 30, 0, 0), fe.Params(2, 1, 2, 1, 12, 3, 5), fe.Params(3, 2, 1, 1, 8, 4, 0)])
def test_prove_verify_parameter_variants(params):
    with fe.field(fe.BABYBEAR):
        o = _system(params, fe.mul_air_inputs())
        packed = fe.pack_claims([])
        proof = o.prove([fe.mul_air_trace(1 << 7)], packed)
        assert o.verify(packed, proof) == 0
        assert o.verify(packed, proof[:-1]) != 0


def test_claims_and_lookups_over_babybear():
    """the even/odd lookup system (src/lookup.rs:975-1007) authored over BabyBear: claims enter the logUp balance"""
    with fe.field(fe.BABYBEAR):
        o = _system(fe.test_params(), fe.even_odd_inputs())
        packed = fe.pack_claims([[0, 4, 1]])
        proof = o.prove(fe.even_odd_traces(), packed)
        assert o.verify(packed, proof) == 0
        assert o.verify(fe.pack_claims([[0, 4, 0]]), proof) != 0
        assert o.verify(fe.pack_claims([]), proof) != 0


@pytest.mark.parametrize("kw", [dict(log_blowup=1, max_log_arity=2), dict(log_blowup=2, max_log_arity=3, commit_proof_of_work_bits=3, cap_height=1),
                                dict(log_blowup=1, max_log_arity=5, log_final_poly_len=1)])
def test_wide_fri_folds_over_babybear(kw):
    """max_log_arity > 1 (src/types.rs:189-190; baby_bear_config.rs:63,79 hands the field to p3-fri): prove -> verify, the step
    arities in the proof, tamper rejection (the fold itself is pinned by tests/test_oracle_fri_arity.py on the other field)"""
    import proof_codec as pc

    params = fe.Params(num_queries=16, **kw)
    with fe.field(fe.BABYBEAR):
        o = _system(params, fe.u32_add_system_inputs())
        traces, claims = fe.u32_add_bench_witness(1 << 7)
        packed = fe.pack_claims(claims)
        proof = o.prove(traces, packed)
        assert o.verify(packed, proof) == 0
        steps = [st["log_arity"] for st in pc.parse(proof, 4, 4)["opening_proof"]["query_proofs"][0]["commit_phase_openings"]]
        assert max(steps) == min(params.max_log_arity, max(steps)) and max(steps) > 1 and len(steps) < 7 + 1 + params.log_blowup
        rng = np.random.default_rng(9)
        rejected = 0
        for _ in range(40):
            bad = bytearray(proof)
            bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
            rejected += o.verify(packed, bytes(bad)) != 0
        assert rejected >= 36  # (a proof-of-work witness is not read at zero bits: a flip there is accepted)
        assert _system(fe.Params(num_queries=16, **dict(kw, max_log_arity=1)), fe.u32_add_system_inputs()).verify(packed, proof) != 0
