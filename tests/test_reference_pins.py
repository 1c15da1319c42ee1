"""The oracle against the REAL reference, field by field - when tests/golden/reference_refs.jsonl exists.

That file is written by the reference itself (bindings/rust/fixture_dump.rs + fixture_hook.patch, one `cargo test` in an
argumentcomputer/multi-stark checkout; INTEGRATION.md "Pinning the oracle"). No Rust toolchain exists where this repository is
built, so the file is absent here and `test_reference_fixture_present` is SKIPPED with the reason "parity unpinned": that
skip IS the parity status. The consumer is nevertheless run in every CPU pass on an emulated file of the same format
(tests/reference_fixtures.py::emulate), so that it works on the day the real file is dropped in."""
import os

import numpy as np
import pytest

import reference_fixtures as rf


def _limbs(d):
    return [int.from_bytes(d[8 * i:8 * i + 8], "little") for i in range(4)]


class Checker:
    """every check of one fixture file; `prover(system_obj, blob, traces, packed)` optionally adds the HIP prover"""

    def __init__(self, lines, fe, oracle, oracle_bb, gpu=None):
        self.lines, self.fe, self.oracle, self.oracle_bb, self.gpu = lines, fe, oracle, oracle_bb, gpu
        self.constants = None
        for ln in lines:
            if ln["kind"] == "babybear_poseidon2":
                self.constants = rf.poseidon2_141(ln)
        self.report = []

    # ---- src/types.rs:246-285
    def pcs_refs(self, ln):
        o = self.oracle
        for n in (3, 17, 22, 20):
            assert _limbs(o.hash_elems(list(range(1, n + 1)))) == ln["LEAF%d" % n], "SerializingHasher<Blake3> leaf of 1..%d" % n
        dig = lambda xs: b"".join(int(x).to_bytes(8, "little") for x in xs)  # noqa: E731
        assert _limbs(o.compress2(dig([1, 2, 3, 4]), dig([5, 6, 7, 8]))) == ln["COMPRESS"], "CompressionFunctionFromHasher"
        m0 = np.zeros((8, 2), dtype=np.uint64)
        m0[5] = [11, 12]
        m1 = np.zeros((4, 3), dtype=np.uint64)
        m1[2] = [107, 108, 109]
        m2 = np.zeros((2, 1), dtype=np.uint64)
        m2[1] = [202]
        t = o.Mmcs([m0, m1, m2])
        vals, proof = t.open(5)
        assert [int(x) for x in vals] == ln["OPENED"], "open_batch(5): opened values"
        assert [_limbs(proof[32 * i:32 * i + 32]) for i in range(len(proof) // 32)] == ln["SIBLINGS"], "open_batch(5): siblings (mixed-height injection)"
        commit = bytes.fromhex(ln["COMMIT_hex"])
        assert commit[-32:] == t.cap and len(commit) in (32, 40), "MerkleTreeMmcs commitment (and its serde form: count + digests)"
        self.report.append("pcs_refs: leaf hash, compress, mixed-height Merkle tree and opening agree")

    # ---- src/types.rs:287-318
    def challenger_refs(self, ln):
        o = self.oracle
        ch = o.Challenger(b"")
        ch.observe(0x0102030405060708)
        assert ch.sample_bits(20) == ln["SAMPLE_BITS"], "sample_bits(20)"
        ch = o.Challenger(b"")
        ch.observe(0x0102030405060708)
        ch.observe(0x1122334455667788)
        assert list(ch.sample_ext()) == ln["APCS"] and list(ch.sample_ext()) == ln["AFRI"], "sample_algebra_element x 2"
        ch.observe(0x00000000DEADBEEF)
        assert list(ch.sample_ext()) == ln["BETA"], "sample after observe"
        ch.observe(0x0A0B0C0D01020304)
        ch.observe(2)
        assert ch.sample_bits(20) == ln["SAMPLE_BITS2"], "sample_bits after two observes"
        self.report.append("challenger_refs: HashChallenger / SerializingChallenger64 sequences agree")

    def babybear_poseidon2(self, ln):
        ob = self.oracle_bb
        ob.set_poseidon2(self.constants)
        got = [int(x) for x in ob.poseidon2_permute(np.arange(16))]
        assert got == ln["permute_0_to_15"], ("Poseidon2BabyBear<16>: the permutation rebuilt from the dumped constants does not reproduce the "
                                              "reference's image of [0..15] (a wrong replay of the RNG stream in fixture_dump.rs, or a wrong oracle)")
        self.report.append("babybear_poseidon2: permutation with the reference's constants agrees")

    # ---- one proof of the reference's suite
    def proof(self, ln):
        fe = self.fe
        bb = rf.is_babybear(ln)
        omod = self.oracle_bb if bb else self.oracle
        name = rf.short_name(ln)
        ref = bytes.fromhex(ln["proof_hex"])
        assert len(ref) == ln["proof_len"]
        claims = rf.claims_of(ln)
        if bb:
            assert self.constants is not None, "a BabyBear case needs the babybear_poseidon2 line of the same file"
            omod.set_poseidon2(self.constants)

        def build(prm):
            if bb:
                with fe.field(fe.BABYBEAR):
                    return fe.system_blob(prm, rf.compiled_circuits(ln, fe), self.constants)
            return fe.system_blob(prm, rf.compiled_circuits(ln, fe))

        with (fe.field(fe.BABYBEAR) if bb else _null()):
            packed = fe.pack_claims(claims)

        def accepts(prm):
            try:
                return omod.System(build(prm)).verify(packed, ref) == 0
            except RuntimeError:
                return False

        prm = rf.params_for(ln, ref, fe, accepts) if name not in ("",) else None
        blob = build(prm)
        osys = omod.System(blob)
        # 1. what System::new derives from the graph
        for ci, c in enumerate(ln["circuits"]):
            info = osys.circuit_info(ci)
            assert (info["stage2_width"], info["constraint_count"], info["max_constraint_degree"]) == (
                c["stage_2_width"], c["constraint_count"], c["max_constraint_degree"]), "%s: circuit %d metadata" % (name, ci)
        # 2. the preprocessed commitment
        pc = osys.preprocessed_commit()
        if ln["preprocessed_commit_hex"] is None:
            assert pc is None
        else:
            assert bytes.fromhex(ln["preprocessed_commit_hex"])[-len(pc):] == pc, "%s: preprocessed commitment" % name
        # 3. the reference's proof is accepted by the oracle's verifier (transcript, commitments, FRI, layout)
        assert osys.verify(packed, ref) == 0, "%s: the oracle's verifier rejects the reference's proof" % name
        # 4. the oracle's prover produces the reference's bytes (serial reference build: the smallest PoW witness)
        traces = rf.traces_of(ln)
        if traces is not None:
            mine = osys.prove(traces, packed)
            if mine != ref:
                first = next((i for i in range(min(len(mine), len(ref))) if mine[i] != ref[i]), min(len(mine), len(ref)))
                raise AssertionError("%s: oracle proof (%d bytes) differs from the reference's (%d bytes) from byte %d" % (name, len(mine), len(ref), first))
            if self.gpu is not None:
                got = self.gpu(bb, blob, len(ln["circuits"]), traces, packed, self.constants)
                assert got == ref, "%s: the HIP prover's proof differs from the reference's" % name
        # 5. the front-end restatement compiles the same graph, where it has the circuit
        with (fe.field(fe.BABYBEAR) if bb else _null()):  # (circuits are authored over the configuration's field)
            mine_inputs = _frontend_inputs(fe, name)
        if mine_inputs is not None:
            with (fe.field(fe.BABYBEAR) if bb else _null()):
                comp = [fe.compile_circuit(ci) for ci in mine_inputs]
                mine_blob = fe.system_blob(prm, comp, self.constants) if bb else fe.system_blob(prm, comp)
            assert mine_blob == blob, "%s: frontend.py compiles a different node vector than the reference's graph::compile" % name
        self.report.append("proof %s: metadata, preprocessed commitment, verifier, %s%s agree" % (
            name, "prover bytes" if traces is not None else "(no traces in the file)", ", front-end graph" if mine_inputs is not None else ""))

    def run(self):
        kinds = {"pcs_refs": self.pcs_refs, "challenger_refs": self.challenger_refs, "babybear_poseidon2": self.babybear_poseidon2,
                 "proof": self.proof}
        for ln in self.lines:
            kinds[ln["kind"]](ln)
        return self.report


class _null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


def _frontend_inputs(fe, name):
    """the circuits of the reference's scenarios that multi-stark_amd/frontend.py restates"""
    if name.startswith("simple_proof"):
        return fe.pythagorean_inputs()
    if name == "u32_add_proof":
        return fe.u32_add_system_inputs()
    if name == "lookup_test":
        return fe.even_odd_inputs()
    if name == "baby_bear_poseidon2_smoke_test":
        return fe.mul_air_inputs()
    if name == "byte_test":
        return fe.byte_operations_inputs()
    if name in ("test_compression_reference_compatibility", "test_all_claims"):  # src/test_circuits/blake3.rs:2215-2613
        import importlib

        return importlib.import_module("multi_stark_amd.blake3_circuit").blake3_system_inputs()
    return None


@pytest.fixture(scope="module")
def oracles():
    import oracle
    import oracle_bb

    oracle.build()
    return oracle, oracle_bb


def test_reference_fixture_present(pkg, oracles):
    """THE parity pin: the reference's own outputs against the oracle. Skipped = parity unpinned."""
    if not os.path.exists(rf.FIXTURE):
        pytest.skip("parity unpinned: tests/golden/reference_refs.jsonl is absent (it is written by the reference itself: "
                    "bindings/rust/fixture_dump.rs, INTEGRATION.md 'Pinning the oracle'; no Rust toolchain exists in this environment)")
    lines = rf.load()
    assert not any(ln.get("emulated") for ln in lines), "tests/golden/reference_refs.jsonl is an emulated file: it pins nothing"
    report = Checker(lines, pkg.frontend, *oracles).run()
    assert any(r.startswith("proof") for r in report) and any(r.startswith("pcs_refs") for r in report)
    print("\n".join(report))


def test_consumer_on_emulated_file(pkg, oracles, tmp_path):
    """the consumer end to end on a file of the reference's format written by the oracle (pins nothing; see the module docs)"""
    path = str(tmp_path / "emulated_refs.jsonl")
    n = rf.emulate(path, oracles[0], oracles[1], pkg.frontend)
    lines = rf.load(path)
    assert len(lines) == n and all(ln["emulated"] for ln in lines)
    report = Checker(lines, pkg.frontend, *oracles).run()
    assert len(report) == n
    # the unknown case had its parameters inferred from the proof, proof-of-work widths included
    unk = [ln for ln in lines if ln.get("test", "").endswith("unknown_case")][0]
    fe = pkg.frontend
    ref = bytes.fromhex(unk["proof_hex"])
    prm = rf.params_for(unk, ref, fe, lambda p: oracles[0].System(fe.system_blob(p, rf.compiled_circuits(unk, fe))).verify(fe.pack_claims([]), ref) == 0)
    assert prm.words() == [1, 1, 1, 1, 9, 2, 3]


def test_consumer_detects_a_divergence(pkg, oracles, tmp_path):
    """a fixture that disagrees with the oracle in one challenge, one digest byte or one proof byte must fail the check"""
    import json

    path = str(tmp_path / "emulated_refs.jsonl")
    rf.emulate(path, oracles[0], oracles[1], pkg.frontend)
    lines = rf.load(path)

    def run_with(mutate):
        import copy

        ls = copy.deepcopy(lines)
        mutate(ls)
        Checker(ls, pkg.frontend, *oracles).run()

    def flip_proof(ls):
        c = [ln for ln in ls if ln["kind"] == "proof" and ln["test"].endswith("lookup_test")][0]
        b = bytearray(bytes.fromhex(c["proof_hex"]))
        b[len(b) // 2] ^= 1
        c["proof_hex"] = bytes(b).hex()

    for mutate in (lambda ls: ls[1].__setitem__("BETA", [1, 2]),
                   lambda ls: ls[0].__setitem__("COMPRESS", [1, 2, 3, 4]),
                   flip_proof):
        with pytest.raises(AssertionError):
            run_with(mutate)
    json.dumps(lines[0])  # (the emulated lines are plain JSON values)


@pytest.mark.gpu
def test_reference_fixture_on_the_hip_prover(pkg, ctx, oracles):
    """with the real file present: the HIP prover reproduces every reference proof whose traces the file holds"""
    if not os.path.exists(rf.FIXTURE):
        pytest.skip("parity unpinned: tests/golden/reference_refs.jsonl is absent")

    def gpu(bb, blob, n, traces, packed, consts):
        if bb:
            s = pkg.babybear.System(ctx, blob, n)
            return s.prove_multiple_claims(s.witness(traces, packed)).to_bytes()
        s = pkg.System(ctx, blob, n)
        return s.prove_multiple_claims(s.witness(traces, packed)).to_bytes()

    Checker(rf.load(), pkg.frontend, *oracles, gpu=gpu).run()
