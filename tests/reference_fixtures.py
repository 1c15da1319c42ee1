"""Consumer of the reference-side fixture file (bindings/rust/fixture_dump.rs -> tests/golden/reference_refs.jsonl): one JSON
object per line, written by the REAL reference (cargo test in an argumentcomputer/multi-stark checkout with the three-line
hook of bindings/rust/fixture_hook.patch). Every object is checked against the oracle field by field; that is what turns
"parity unpinned" into "pinned" for commitments, challenges, FRI contents and Proof::to_bytes.

`emulate(path)` writes a file of the SAME format from the oracle itself. It exists so that the consumer below is exercised
in every CPU run although no reference output can be produced in this environment (no cargo): a consumer that had never
run would be of little use on the day the real file arrives. An emulated file pins nothing and says so (`"emulated": true`
on every line); tests/test_reference_pins.py never looks for it under tests/golden/.
Test infrastructure only."""
import json
import os

import numpy as np

import proof_codec

BB_P = (1 << 31) - (1 << 27) + 1
BB_R_INV = pow(1 << 32, BB_P - 2, BB_P)  # serde of MontyField31 = the Montgomery word x * 2^32 mod p
FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_refs.jsonl")

# parameters of the reference's own scenarios, by the test that produces them ([log_blowup, cap_height, log_final_poly_len,
# max_log_arity, num_queries, commit_pow_bits, query_pow_bits]); anything else is inferred from the proof (infer_params)
TEST_PARAMS = [1, 0, 0, 1, 64, 0, 0]
KNOWN_PARAMS = {
    "simple_proof_bench_params": [2, 0, 0, 1, 100, 10, 10],   # benches/multi_stark.rs:244-258
    "simple_proof_cap2_final4": [2, 2, 2, 1, 20, 3, 5],
    "simple_proof_arity4": [1, 0, 0, 2, 20, 0, 0],            # FRI rounds of arity 4 / 8 (src/types.rs:189-190): no call site of the
    "simple_proof_arity8_pow": [2, 1, 1, 3, 20, 3, 5],        # reference sets them; these two pin the restated wide rounds
}


def load(path=FIXTURE):
    out = []
    with open(path) as f:
        for n, line in enumerate(f, 1):
            line = line.strip()
            if not line:
                continue
            try:
                out.append(json.loads(line))
            except ValueError as e:
                raise ValueError("%s line %d is not JSON: %s" % (path, n, e))
    return out


def short_name(case):
    return case.get("test", "?").split("::")[-1]


def is_babybear(case):
    return case.get("elem_bytes", 8) == 4


def canon(case, words):
    """serde words -> canonical field elements (identity for Goldilocks, out of Montgomery form for BabyBear)"""
    a = np.asarray(words, dtype=np.uint64)
    if not is_babybear(case):
        return a
    return np.asarray([(int(x) * BB_R_INV) % BB_P for x in a.reshape(-1)], dtype=np.uint64).reshape(a.shape)


def compiled_circuits(case, fe):
    """the reference's compiled graphs (src/graph.rs:62-76) as the front-end's CompiledCircuit objects"""
    out = []
    for c in case["circuits"]:
        nodes = []
        for kind, a, b, source, offset in c["nodes"]:
            if kind == 0:  # Const: the only node that carries a field element
                a = int(canon(case, [a])[0])
            nodes.append((kind, source, offset, int(a), int(b)))
        pre = None
        if c["preprocessed"] is not None:
            pre = canon(case, c["preprocessed"]).reshape(c["preprocessed_height"], c["preprocessed_width"])
        cc = fe.CompiledCircuit(nodes, [int(z) for z in c["zeros"]], [(int(m), [int(x) for x in args]) for m, args in c["lookups"]],
                                c["main_width"], pre)
        cc.lookup_prefix_len = c.get("lookup_prefix_len", 0)
        out.append(cc)
    return out


def traces_of(case):
    if case.get("traces") is None:
        return None
    return [canon(case, t["values"]).reshape(t["height"], t["width"]) if t["height"] else np.zeros((0, max(t["width"], 1)), dtype=np.uint64)
            for t in case["traces"]]


def claims_of(case):
    return [[int(x) for x in canon(case, c)] if len(c) else [] for c in case["claims"]]


def infer_params(case, proof):
    """[log_blowup, cap_height, log_final_poly_len, max_log_arity, num_queries] from the proof's own shape; the two
    proof-of-work widths cannot be read off a proof (None, None): the caller tries them (the seed binds them, so a wrong
    guess is rejected)"""
    bb = is_babybear(case)
    try:
        p = proof_codec.parse(proof, 4 if bb else 8, 4 if bb else 2)
    except (ValueError, IndexError, AssertionError) as e:
        raise AssertionError("%s: the reference's Proof::to_bytes does not parse under the restated layout (tests/proof_codec.py): %s" % (
            case.get("test"), e))
    fri = p["opening_proof"]
    arities = {o["log_arity"] for q in fri["query_proofs"] for o in q["commit_phase_openings"]}
    return [case["log_blowup"], len(p["stage_1_commit"]).bit_length() - 1, len(fri["final_poly"]).bit_length() - 1,
            max(arities) if arities else 1, len(fri["query_proofs"]), None, None], p


def params_for(case, proof, fe, verify):
    """the Params of a case: by name, else from the proof's shape plus a search over the proof-of-work widths with `verify(params)`"""
    name = short_name(case)
    if name in KNOWN_PARAMS:
        return fe.Params(*KNOWN_PARAMS[name])
    shape, parsed = infer_params(case, proof)
    fri = parsed["opening_proof"]
    zero_pow = all(w == 0 for w in fri["commit_pow_witnesses"]) and fri["query_pow_witness"] == 0
    # all-zero witnesses mean zero proof-of-work bits (DeterministicPow, src/types.rs:75-80; a nonzero width leaves an all-zero
    # set of witnesses with negligible probability); otherwise every pair of widths up to 20 bits is tried
    cands = [(0, 0)] if zero_pow else [(c, q) for c in range(0, 21) for q in range(0, 21) if (c, q) != (0, 0)]
    for c, q in cands:
        prm = fe.Params(*(shape[:5] + [c, q]))
        if verify(prm):
            return prm
    raise AssertionError("no proof-of-work widths in 0..20 make the oracle accept the reference proof of %s (shape %s): the transcript, "
                         "the commitments or the proof layout differ" % (case.get("test"), shape[:5]))


def poseidon2_141(line):
    """the 141 constants in the front-end's order: 8 external rounds x 16 (initial then terminal), then 13 internal"""
    k = list(line["external_initial"]) + list(line["external_terminal"]) + list(line["internal"])
    assert len(k) == 141, "expected 8 x 16 + 13 Poseidon2 constants, got %d" % len(k)
    return np.asarray(k, dtype=np.uint64)


# ---------------------------------------------------------------------------------------------------------------------
def _words(case_bb, arr):
    a = np.asarray(arr, dtype=np.uint64).reshape(-1)
    if not case_bb:
        return [int(x) for x in a]
    return [(int(x) << 32) % BB_P for x in a]


def emulate(path, oracle, oracle_bb, fe):
    """Write an emulated fixture file (format of bindings/rust/fixture_dump.rs) from the oracle: see the module docstring."""
    lines = []

    def limbs(d):
        return [int.from_bytes(d[8 * i:8 * i + 8], "little") for i in range(4)]

    # pcs_refs (src/types.rs:246-285)
    o = {"kind": "pcs_refs", "emulated": True}
    for n in (3, 17, 22, 20):
        o["LEAF%d" % n] = limbs(oracle.hash_elems(list(range(1, n + 1))))
    dig = lambda xs: b"".join(int(x).to_bytes(8, "little") for x in xs)  # noqa: E731
    o["COMPRESS"] = limbs(oracle.compress2(dig([1, 2, 3, 4]), dig([5, 6, 7, 8])))
    m0 = np.zeros((8, 2), dtype=np.uint64)
    m0[5] = [11, 12]
    m1 = np.zeros((4, 3), dtype=np.uint64)
    m1[2] = [107, 108, 109]
    m2 = np.zeros((2, 1), dtype=np.uint64)
    m2[1] = [202]
    t = oracle.Mmcs([m0, m1, m2])
    vals, proof = t.open(5)
    o["OPENED"] = [int(x) for x in vals]
    o["SIBLINGS"] = [limbs(proof[32 * i:32 * i + 32]) for i in range(len(proof) // 32)]
    o["COMMIT_hex"] = t.cap.hex()
    lines.append(o)
    # challenger_refs (src/types.rs:287-318)
    ch = oracle.Challenger(b"")
    ch.observe(0x0102030405060708)
    o = {"kind": "challenger_refs", "emulated": True, "SAMPLE_BITS": ch.sample_bits(20)}
    ch = oracle.Challenger(b"")
    ch.observe(0x0102030405060708)
    ch.observe(0x1122334455667788)
    o["APCS"], o["AFRI"] = list(ch.sample_ext()), list(ch.sample_ext())
    ch.observe(0x00000000DEADBEEF)
    o["BETA"] = list(ch.sample_ext())
    ch.observe(0x0A0B0C0D01020304)
    ch.observe(2)
    o["SAMPLE_BITS2"] = ch.sample_bits(20)
    lines.append(o)

    def proof_case(test, config, inputs, params, traces, claims, omod, bb=False, consts=None):
        comp = [fe.compile_circuit(ci) for ci in inputs]
        blob = fe.system_blob(params, comp, consts) if bb else fe.system_blob(params, comp)
        s = omod.System(blob)
        packed = fe.pack_claims(claims)
        p = s.prove(traces, packed)
        circuits = []
        for ci, c in enumerate(comp):
            info = s.circuit_info(ci)
            circuits.append({
                "main_width": c.main_width, "preprocessed_width": info["pre_width"], "preprocessed_height": info["pre_height"],
                "num_lookups": info["num_lookups"], "stage_2_width": info["stage2_width"], "constraint_count": info["constraint_count"],
                "max_constraint_degree": info["max_constraint_degree"], "lookup_prefix_len": c.lookup_prefix_len,
                "nodes": [[k, _words(bb, [a])[0] if k == 0 else a, b, src, off] for (k, src, off, a, b) in c.nodes],
                "zeros": list(c.zeros), "lookups": [[m, list(args)] for m, args in c.lookups],
                "preprocessed": None if c.preprocessed is None else _words(bb, c.preprocessed)})
        pc = s.preprocessed_commit()
        lines.append({"kind": "proof", "emulated": True, "test": test, "config": config, "log_blowup": params.log_blowup,
                      "elem_bytes": 4 if bb else 8, "circuits": circuits,
                      # bincode of MerkleCap<_, [u8; 32]>: length prefix, then the digests
                      "preprocessed_commit_hex": None if pc is None else (len(pc) // 32).to_bytes(8, "little").hex() + pc.hex(),
                      "traces": [{"height": int(t.shape[0]), "width": int(t.shape[1]), "values": _words(bb, t)} for t in traces],
                      "trace_heights": [int(t.shape[0]) for t in traces], "claims": [_words(bb, c) for c in claims],
                      "proof_len": len(p), "proof_hex": p.hex()})

    g = "multi_stark::types::GoldilocksBlake3Config"
    proof_case("fixture_dump::tests::simple_proof_4_rows", g, fe.pythagorean_inputs(), fe.test_params(), [fe.pythagorean_trace(4)], [], oracle)
    proof_case("fixture_dump::tests::simple_proof_bench_params", g, fe.pythagorean_inputs(), fe.Params(*KNOWN_PARAMS["simple_proof_bench_params"]),
               [fe.pythagorean_trace(1024)], [], oracle)
    proof_case("fixture_dump::tests::simple_proof_cap2_final4", g, fe.pythagorean_inputs(), fe.Params(*KNOWN_PARAMS["simple_proof_cap2_final4"]),
               [fe.pythagorean_trace(256)], [], oracle)
    proof_case("fixture_dump::tests::simple_proof_arity4", g, fe.pythagorean_inputs(), fe.Params(*KNOWN_PARAMS["simple_proof_arity4"]),
               [fe.pythagorean_trace(256)], [], oracle)
    proof_case("fixture_dump::tests::simple_proof_arity8_pow", g, fe.pythagorean_inputs(), fe.Params(*KNOWN_PARAMS["simple_proof_arity8_pow"]),
               [fe.pythagorean_trace(512)], [], oracle)
    proof_case("lookup::tests::lookup_test", g, fe.even_odd_inputs(), fe.test_params(), fe.even_odd_traces(), [[0, 4, 1]], oracle)
    tr, cl = fe.u32_add_witness([(10, 5), (30, 20), (100, 100), (8000, 10000)])
    proof_case("test_circuits::u32_add::tests::u32_add_proof", g, fe.u32_add_system_inputs(), fe.test_params(), tr, [list(map(int, c)) for c in cl], oracle)
    # the reference's largest scenario (src/test_circuits/blake3.rs:2215-2340): nine circuits, one compression claim
    import importlib

    b3 = importlib.import_module("multi_stark_amd.blake3_circuit")
    b3_claims = [b3.compression_claim(b3.blake3_compressions(bytes([0x54] * 64))[0][0])]
    proof_case("test_circuits::blake3::tests::test_compression_reference_compatibility", g, b3.blake3_system_inputs(), fe.test_params(),
               b3.blake3_witness(b3_claims), b3_claims, oracle)
    # an unknown test with proof-of-work: the consumer has to infer every parameter
    proof_case("somewhere::else::unknown_case", g, fe.pythagorean_inputs(), fe.Params(1, 1, 1, 1, 9, 2, 3), [fe.pythagorean_trace(64)], [], oracle)
    # BabyBear / Poseidon2: the constants line, then the reference's smoke test (baby_bear_config.rs:159-206)
    k = fe.poseidon2_constants()
    oracle_bb.set_poseidon2(k)
    kk = [int(x) for x in np.asarray(k).reshape(-1)]
    lines.append({"kind": "babybear_poseidon2", "emulated": True, "external_initial": kk[:64], "external_terminal": kk[64:128], "internal": kk[128:],
                  "permute_0_to_15": [int(x) for x in oracle_bb.poseidon2_permute(np.arange(16))]})
    with fe.field(fe.BABYBEAR):
        proof_case("test_circuits::baby_bear_config::baby_bear_poseidon2_smoke_test", "multi_stark::test_circuits::baby_bear_config::BabyBearPoseidon2Config",
                   fe.mul_air_inputs(), fe.test_params(), [fe.mul_air_smoke_trace()], [], oracle_bb, bb=True, consts=k)
    with open(path, "w") as f:
        for o in lines:
            f.write(json.dumps(o) + "\n")
    return len(lines)
