"""CPU test: what the front-end COMPILES means what was AUTHORED. Both sides of every parity test (HIP library and oracle) consume
the node vectors `frontend.compile_circuit` emits, so a folding or interning rule that changed a value would prove and verify
identically on both sides. Here every constraint, multiplicity and lookup argument is evaluated twice on random rows - as the
authored `Expr` tree (plain recursive arithmetic mod p, no folding beyond what the operators did) and as its root in the compiled
node vector (evaluated node by node the way the kernels walk it) - for the reference's circuits incl. the nine of its BLAKE3
system (src/expr.rs:179-227, src/graph.rs:120-324)."""
import importlib

import numpy as np
import pytest


def _eval_expr(fe, e, env, memo):
    k = id(e)
    if k in memo:
        return memo[k]
    P = fe.P
    if e.kind == fe.N_CONST:
        v = e.a % P
    elif e.kind == fe.N_VAR:
        v = int(env["rows"][(e.source, e.offset)][e.a])
    elif e.kind == fe.N_PUBLIC:
        v = env["publics"][e.a]
    elif e.kind == fe.N_IS_FIRST:
        v = env["sel"][0]
    elif e.kind == fe.N_IS_LAST:
        v = env["sel"][1]
    elif e.kind == fe.N_IS_TRANS:
        v = env["sel"][2]
    elif e.kind == fe.N_NEG:
        v = (-_eval_expr(fe, e.a, env, memo)) % P
    else:
        a, b = _eval_expr(fe, e.a, env, memo), _eval_expr(fe, e.b, env, memo)
        v = (a + b) % P if e.kind == fe.N_ADD else (a - b) % P if e.kind == fe.N_SUB else (a * b) % P
    memo[k] = v
    return v


def _eval_nodes(fe, nodes, env):
    P = fe.P
    out = []
    for (kind, source, offset, a, b) in nodes:
        if kind == fe.N_CONST:
            v = a % P
        elif kind == fe.N_VAR:
            v = int(env["rows"][(source, offset)][a])
        elif kind == fe.N_PUBLIC:
            v = env["publics"][a]
        elif kind == fe.N_IS_FIRST:
            v = env["sel"][0]
        elif kind == fe.N_IS_LAST:
            v = env["sel"][1]
        elif kind == fe.N_IS_TRANS:
            v = env["sel"][2]
        elif kind == fe.N_NEG:
            v = (-out[a]) % P
        elif kind == fe.N_ADD:
            v = (out[a] + out[b]) % P
        elif kind == fe.N_SUB:
            v = (out[a] - out[b]) % P
        else:
            v = (out[a] * out[b]) % P
        out.append(v)
    return out


def _systems(fe):
    b3 = importlib.import_module("multi_stark_amd.blake3_circuit")
    return {"pythagorean": fe.pythagorean_inputs, "u32 add + byte table": fe.u32_add_system_inputs, "even / odd lookups": fe.even_odd_inputs,
            "byte operations": fe.byte_operations_inputs, "squares": fe.squares_inputs, "verifier test": fe.verifier_test_inputs,
            "blake3 compression system": b3.blake3_system_inputs}


@pytest.mark.parametrize("name", ["pythagorean", "u32 add + byte table", "even / odd lookups", "byte operations", "squares", "verifier test",
                                  "blake3 compression system"])
def test_compiled_nodes_mean_what_was_authored(fe, name):
    rng = np.random.default_rng(sum(name.encode()))
    P = fe.P
    for ci, inputs in enumerate(_systems(fe)[name]()):
        cc = fe.compile_circuit(inputs)
        # the roots, by compiling again in compile_circuit's order (lookups, then constraints) on a fresh interner
        it = fe._Interner()
        pw = 0 if inputs.preprocessed is None else int(inputs.preprocessed.shape[1])
        spec = {"main_width": inputs.main_width, "preprocessed_width": pw, "stage2_width": max(len(inputs.lookups), 1) * fe.EXT_D, "num_publics": 4 * fe.EXT_D}
        lk_roots = [(it.compile_expr(l.multiplicity, spec, False), [it.compile_expr(a, spec, False) for a in l.args]) for l in inputs.lookups]
        c_roots = [it.compile_expr(c, spec, False) for c in inputs.constraints]
        assert it.nodes[:len(cc.nodes)] == cc.nodes and [(m, a) for m, a in lk_roots] == [(m, list(a)) for m, a in cc.lookups], (name, ci)
        for trial in range(3):
            env = {"rows": {(s, o): [int(x) for x in rng.integers(0, P, max(w, 1), dtype=np.uint64)]
                            for s, w in ((fe.SRC_PRE, pw), (fe.SRC_MAIN, inputs.main_width)) for o in (0, 1)},
                   "publics": [int(x) for x in rng.integers(0, P, 4 * fe.EXT_D, dtype=np.uint64)],
                   "sel": [int(x) for x in rng.integers(0, P, 3, dtype=np.uint64)]}
            vals = _eval_nodes(fe, it.nodes, env)
            memo = {}
            for l, (m, args) in zip(inputs.lookups, lk_roots):
                assert _eval_expr(fe, l.multiplicity, env, memo) == vals[m]
                for a_expr, a_root in zip(l.args, args):
                    assert _eval_expr(fe, a_expr, env, memo) == vals[a_root]
            kept = set(cc.zeros)
            for c_expr, root in zip(inputs.constraints, c_roots):
                assert _eval_expr(fe, c_expr, env, memo) == vals[root], (name, ci)
                # a constraint that folded to the constant zero is dropped; every other root is among the circuit's zeros
                assert root in kept or it.as_const(root) == 0
