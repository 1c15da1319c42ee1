"""CPU tests: the circuit front-end mirrors src/expr.rs folding and src/graph.rs interning."""
import numpy as np
import pytest


def test_expr_constant_folding(fe):
    E = fe.Expr
    x = E.main(0)
    assert (E.const(2) + E.const(3)).a == 5
    assert (x + 0) is x and (0 + x) is x and (x * 1) is x and (x - 0) is x
    assert (x * 0).is_const(0)
    assert (0 - x).kind == fe.N_NEG and (-(-x)) is x
    assert (E.const(0) - E.const(1)).a == fe.P - 1


def test_interning_and_canonical_zeros(fe):
    E = fe.Expr
    a, b = E.main(0), E.main(1)
    ci = fe.CircuitInputs(2, None, [a + b, b + a, (a + b) - (a + b), a * b - b * a + a], [], [])
    cc = fe.compile_circuit(ci)
    # a+b and b+a intern to one node; x - x folds to the zero constant and is dropped; a*b - b*a folds to 0 so "+ a" is a
    assert len(cc.zeros) == 2
    kinds = [n[0] for n in cc.nodes]
    assert kinds.count(fe.N_ADD) == 1
    with pytest.raises(fe.CompileError):
        fe.compile_circuit(fe.CircuitInputs(1, None, [E.const(5)], [], []))
    with pytest.raises(fe.CompileError):
        fe.compile_circuit(fe.CircuitInputs(1, None, [E.main(3)], [], []))


def test_u32_add_program_shape(fe, oracle):
    comp = [fe.compile_circuit(ci) for ci in fe.u32_add_system_inputs()]
    byte, add = comp
    assert len(byte.zeros) == 0 and len(byte.lookups) == 1
    assert len(add.zeros) == 2 and len(add.lookups) == 13
    assert [len(a) for _, a in add.lookups] == [4] + [2] * 12
    # children precede parents
    for i, (kind, _, _, a, b) in enumerate(add.nodes):
        if kind in (fe.N_ADD, fe.N_SUB, fe.N_MUL):
            assert a < i and b < i
    s = oracle.System(fe.system_blob(fe.bench_params(), comp))
    # benches/multi_stark.rs system: widths / constraint counts
    assert s.circuit_info(0) == {"main_width": 1, "pre_width": 1, "pre_height": 256, "num_lookups": 1, "stage2_width": 2,
                                 "constraint_count": 2, "max_constraint_degree": 2, "quotient_degree": 1, "args_width": 2}
    assert s.circuit_info(1) == {"main_width": 14, "pre_width": 0, "pre_height": 0, "num_lookups": 13, "stage2_width": 26,
                                 "constraint_count": 28, "max_constraint_degree": 2, "quotient_degree": 1, "args_width": 28}


def test_ext_constraints_karatsuba(fe):
    E, X = fe.Expr, fe.ExtExpr
    a = X.coords([E.main(0), E.main(1)])
    b = X.coords([E.main(2), E.main(3)])
    cc = fe.compile_circuit(fe.CircuitInputs(4, None, [], [a * b - X.coords([E.main(0), E.main(1)])], []))
    assert len(cc.zeros) == 2
    assert [n[0] for n in cc.nodes].count(fe.N_MUL) == 4  # 3 Karatsuba products + W * p1
    with pytest.raises(fe.CompileError):
        fe.compile_circuit(fe.CircuitInputs(1, None, [], [X.base(E.main(0)) * X.base(E.main(0))], []))


def test_bench_witness_matches_reference_generator(fe):
    traces, claims = fe.u32_add_bench_witness(8)
    # first xorshift32 outputs from 0xdeadbeef / 0xcafebabe (benches/multi_stark.rs:180-192)
    a = 0xDEADBEEF
    a ^= (a << 13) & 0xFFFFFFFF
    a ^= a >> 17
    a ^= (a << 5) & 0xFFFFFFFF
    assert int(claims[0, 1]) == a
    assert int(claims[0, 0]) == 1 and int(claims[0, 3]) == (int(claims[0, 1]) + int(claims[0, 2])) & 0xFFFFFFFF
    byte, add = traces
    assert byte.shape == (256, 1) and add.shape == (8, 14) and int(byte.sum()) == 12 * 8
    x = sum(int(add[0, k]) << (8 * k) for k in range(4))
    assert x == int(claims[0, 1]) and int(add[0, 13]) == 1


def test_node_vectors_have_not_drifted(fe):
    """tests/golden/frontend_nodes.json records the node vectors this front-end compiles for the reference's circuits. Both
    the HIP library and the oracle consume its output, so a change here (interning order, folding rule, lookup prefix) would
    prove and verify identically on both sides: this test is what notices. (Correctness against the reference's
    graph::compile is checked by tests/test_reference_pins.py when the reference's fixture file is present.)"""
    import importlib.util
    import json
    import os

    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_frontend_golden", os.path.join(here, "make_frontend_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = json.load(open(os.path.join(here, "frontend_nodes.json")))["circuits"]
    got = json.loads(json.dumps(mod.describe(fe)))
    assert set(got) == set(want)
    for name in want:
        assert got[name] == want[name], "front-end output changed for: %s (regenerate the golden file only if the change is intended)" % name


def test_small_rng_restatement_matches_published_vectors(fe):
    """The generator behind the BabyBear configuration's Poseidon2 constants (SmallRng::seed_from_u64(42),
    src/test_circuits/baby_bear_config.rs:54-55) is xoshiro256++ seeded through SplitMix64: both are published algorithms with
    published vectors (xoshiro256plusplus.c from state {1, 2, 3, 4}; SplitMix64 from 0). What stays UPSTREAM-RECALL is how rand
    and p3 wire them together (frontend.poseidon2_constants_small_rng) - the pinning kit's 141 dumped constants settle that."""
    r = fe.SmallRngXoshiro.__new__(fe.SmallRngXoshiro)
    r.s = [1, 2, 3, 4]
    assert [r.next_u64() for _ in range(10)] == [
        41943041, 58720359, 3588806011781223, 3591011842654386, 9228616714210784205, 9973669472204895162, 14011001112246962877,
        12406186145184390807, 15849039046786891736, 10450023813501588000]
    assert fe.SmallRngXoshiro(0).s == [0xE220A8397B1DCDAF, 0x6E789E6AA1B965F4, 0x06C45D188009454F, 0xF88BB8A8724C81EC]
    k = fe.poseidon2_constants()
    assert len(k) == 141 and int(k.max()) < fe.BABYBEAR["P"] and len(set(int(x) for x in k)) == 141
    assert (k == fe.poseidon2_constants_small_rng(42)).all() and not (k == fe.poseidon2_constants_stand_in(42)).all()
    # the Montgomery word of the first constant is the first accepted 31-bit draw
    first = fe.SmallRngXoshiro(42)
    while True:
        x = first.next_u32() >> 1
        if x < fe.BABYBEAR["P"]:
            break
    assert int(k[0]) * (1 << 32) % fe.BABYBEAR["P"] == x
