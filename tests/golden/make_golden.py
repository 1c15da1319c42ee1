"""Regenerates tests/golden/*.json from the ORACLE (oracle/libms_oracle.so).

SELF-GENERATED, NOT REFERENCE-VERIFIED: the reference (Rust + Plonky3) cannot be built or run in this environment
and holds no literal vectors for these values (its gen_pcs_refs / gen_challenger_refs tests only print,
/root/reference/src/types.rs:246-319). The scenarios below are exactly those named scenarios plus the reference's
end-to-end test inputs, so that a later session with a Rust toolchain can diff the outputs.
Run: python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402
import oracle  # noqa: E402


def limbs(d):
    return [int.from_bytes(d[8 * i:8 * i + 8], "little") for i in range(4)]


def pcs_refs():
    out = {}
    for n in (3, 17, 22, 20):  # src/types.rs:250-253
        out["LEAF%d" % n] = limbs(oracle.hash_elems(list(range(1, n + 1))))
    dig = lambda xs: b"".join(int(x).to_bytes(8, "little") for x in xs)  # noqa: E731
    out["COMPRESS"] = limbs(oracle.compress2(dig([1, 2, 3, 4]), dig([5, 6, 7, 8])))
    m0 = np.zeros((8, 2), dtype=np.uint64)
    m0[5] = [11, 12]
    m1 = np.zeros((4, 3), dtype=np.uint64)
    m1[2] = [107, 108, 109]
    m2 = np.zeros((2, 1), dtype=np.uint64)
    m2[1] = [202]
    t = oracle.Mmcs([m0, m1, m2])
    vals, proof = t.open(5)
    out["OPENED"] = [int(x) for x in vals]
    out["SIBLINGS"] = [limbs(proof[32 * i:32 * i + 32]) for i in range(len(proof) // 32)]
    out["COMMIT"] = limbs(t.cap)
    return out


def challenger_refs():
    out = {}
    ch = oracle.Challenger(b"")  # src/types.rs:296-301
    ch.observe(0x0102030405060708)
    out["SAMPLE_BITS"] = ch.sample_bits(20)
    ch = oracle.Challenger(b"")  # src/types.rs:303-318
    ch.observe(0x0102030405060708)
    ch.observe(0x1122334455667788)
    out["APCS"] = list(ch.sample_ext())
    out["AFRI"] = list(ch.sample_ext())
    ch.observe(0x00000000DEADBEEF)
    out["BETA"] = list(ch.sample_ext())
    ch.observe(0x0A0B0C0D01020304)
    ch.observe(2)
    out["SAMPLE_BITS2"] = ch.sample_bits(20)
    return out


def proofs():
    fe = load_package().frontend
    out = {}

    def run(name, inputs, params, traces, claims):
        comp = [fe.compile_circuit(ci) for ci in inputs]
        blob = fe.system_blob(params, comp)
        s = oracle.System(blob)
        packed = fe.pack_claims(claims)
        p = s.prove(traces, packed)
        assert s.verify(packed, p) == 0
        out[name] = {"blob_sha256": hashlib.sha256(blob).hexdigest(), "proof_len": len(p),
                     "proof_sha256": hashlib.sha256(p).hexdigest(), "stage1_commit": p[8 + len(comp) + 8: 8 + len(comp) + 40].hex()}

    run("simple_proof_4", fe.pythagorean_inputs(), fe.test_params(), [fe.pythagorean_trace(4)], [])
    run("simple_proof_4096", fe.pythagorean_inputs(), fe.test_params(), [fe.pythagorean_trace(4096)], [])
    run("lookup_even_odd", fe.even_odd_inputs(), fe.test_params(), fe.even_odd_traces(), [[0, 4, 1]])
    tr, cl = fe.u32_add_witness([(10, 5), (30, 20), (100, 100), (8000, 10000)])
    run("u32_add_proof", fe.u32_add_system_inputs(), fe.test_params(), tr, cl)
    tr, cl = fe.u32_add_bench_witness(1 << 8)
    run("bench_u32_add_2p8", fe.u32_add_system_inputs(), fe.bench_params(), tr, cl)
    return out


def babybear_refs():
    """The reference's second configuration (src/test_circuits/baby_bear_config.rs) on oracle/libms_oracle_bb.so, instantiated
    as the reference does - Perm::new_from_rng_128(SmallRng::seed_from_u64(42)), restated in frontend.poseidon2_constants_small_rng
    (UPSTREAM-RECALL: checked by the pinning kit's dumped constants when they arrive)."""
    import oracle_bb as ob

    fe = load_package().frontend
    k = fe.poseidon2_constants()
    ob.set_poseidon2(k)
    out = {"constants_sha256": hashlib.sha256(np.ascontiguousarray(k, dtype=np.uint64).tobytes()).hexdigest(),
           "permute_0_to_15": [int(x) for x in ob.poseidon2_permute(np.arange(16))],
           "leaf_1_to_11": ob.hash_elems(list(range(1, 12))).hex()}
    ch = ob.Challenger(b"multi-stark/v0")
    ch.observe(7)
    out["challenger_sample_ext"] = list(ch.sample_ext())
    out["challenger_sample_bits_20"] = ch.sample_bits(20)
    with fe.field(fe.BABYBEAR):
        comp = [fe.compile_circuit(c) for c in fe.mul_air_inputs()]
        blob = fe.system_blob(fe.test_params(), comp, k)
        s = ob.System(blob)
        packed = fe.pack_claims([])
        p = s.prove([fe.mul_air_smoke_trace()], packed)  # baby_bear_config.rs:159-206
        assert s.verify(packed, p) == 0
        out["mul_air_smoke_test"] = {"blob_sha256": hashlib.sha256(blob).hexdigest(), "proof_len": len(p),
                                     "proof_sha256": hashlib.sha256(p).hexdigest()}
        p = s.prove([fe.mul_air_trace(1 << 10)], packed)
        out["mul_air_1024"] = {"proof_len": len(p), "proof_sha256": hashlib.sha256(p).hexdigest()}
    return out


if __name__ == "__main__":
    data = {"_note": "SELF-GENERATED by the oracle, not reference-verified (see module docstring)",
            "pcs_refs": pcs_refs(), "challenger_refs": challenger_refs(), "proofs": proofs(), "babybear": babybear_refs()}
    with open(os.path.join(HERE, "oracle_refs.json"), "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(HERE, "oracle_refs.json"))
