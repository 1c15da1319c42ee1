"""Differential fuzzing of the whole path: random systems (random constraint graphs, lookups, preprocessed traces,
heights, inactive circuits, ragged claims, PCS/FRI parameters) proved by the HIP library and by the oracle; the proof
bytes must be identical and the two verifiers must agree. The witnesses are random, so most proofs do not verify - that
is irrelevant for parity: both provers must still emit the same bytes, and both verifiers the same verdict class.

usage: python3 tools/fuzz_parity.py [N_CASES] [SEED]      (MSAMD_NO_JIT=1 skips the hiprtc compile of every new circuit)
       FUZZ_BIG=1 ... wider and taller systems;  FUZZ_CLAIMS=1 ... more than 8192 claim words (device-side outer transcript)
       FUZZ_MANY=1 ... systems of 4 .. 40 circuits
       FUZZ_ARITY=1 ... FriParameters::max_log_arity drawn from 1..6 (FRI rounds of arity up to 64)
       FUZZ_LEVEL2=1 ... every proved case also through the Level-2 entry points (the prover loop of tests/test_gpu_level2.py /
                         test_gpu_bb_level2.py, one device call per step): same bytes
       FUZZ_FAULTS=1 ... every proved case also with a device allocation failing at three random points of the proof
                         (ms_ctx_debug_fail_alloc): an error with the injected reason or the same bytes, then the same bytes again
       FUZZ_PARAMS=1 ... wider protocol parameters (caps up to 2^6 digests, final polynomials up to 2^5 coefficients, up to 120
                         queries, up to 12 + 12 proof-of-work bits) and every third case also from a host-resident witness
       FUZZ_FIELD=babybear python3 tools/fuzz_parity.py ...   the same systems over the reference's second configuration
       (BabyBear / Poseidon2, include/mstark_bb.h) against oracle/libms_oracle_bb.so
The oracle is used only as the checker."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402

BABYBEAR = os.environ.get("FUZZ_FIELD", "") == "babybear"
P = 2013265921 if BABYBEAR else 0xFFFFFFFF00000001
KPERM = None
BIG = bool(os.environ.get("FUZZ_BIG"))  # wider / taller systems: rows over one BLAKE3 chunk, > 16 lookups, 2^15 rows
LONG_CLAIMS = bool(os.environ.get("FUZZ_CLAIMS"))  # more than 8192 claim words: the claims digest and the outer transcript run on the device


def rand_field(rng, shape):
    v = rng.integers(0, P, shape, dtype=np.uint64)
    edge = np.array([0, 1, 2, P - 1, P - 2, (1 << 32) - 1, 1 << 32, (1 << 32) + 1], dtype=np.uint64) % np.uint64(P)
    mask = rng.random(shape) < 0.15
    return np.where(mask, edge[rng.integers(0, len(edge), shape)], v)


def random_expr(rng, fe, atoms, depth, max_degree):
    """(expr, degree) with degree <= max_degree"""
    E = fe.Expr
    if depth == 0 or rng.random() < 0.25:
        k = rng.integers(0, len(atoms) + 1)
        if k == len(atoms):
            return E.const(int(rng.integers(0, P, dtype=np.uint64))), 0
        return atoms[k], 1
    op = rng.integers(0, 4)
    a, da = random_expr(rng, fe, atoms, depth - 1, max_degree)
    if op == 3:
        return -a, da
    b, db = random_expr(rng, fe, atoms, depth - 1, max_degree)
    if op == 2:
        if da + db <= max_degree:
            return a * b, da + db
        return a + b, max(da, db)
    return (a + b, max(da, db)) if op == 0 else (a - b, max(da, db))


def random_circuit(rng, fe, log_blowup):
    E = fe.Expr
    w = int(rng.integers(1, 150 if BIG and rng.random() < 0.3 else 7))
    pw = int(rng.integers(0, 4)) if rng.random() < 0.4 else 0
    h = 1 << int(rng.integers(0, 16 if BIG else 11))
    pre = rand_field(rng, (h, pw)) if pw else None
    atoms = [E.main(i) for i in range(w)] + [E.main_next(i) for i in range(w)]
    atoms += [E.var(fe.SRC_PRE, 0, i) for i in range(pw)] + [E.var(fe.SRC_PRE, 1, i) for i in range(pw)]
    max_deg = min(3, (1 << log_blowup) + 1)

    def ev(b):
        for _ in range(int(rng.integers(0, 4))):
            e, _d = random_expr(rng, fe, atoms, 3, max_deg - (1 if rng.random() < 0.3 else 0))
            r = rng.random()
            if r < 0.2:
                b.when_transition().assert_zero(e) if _d < max_deg else b.assert_zero(e)
            elif r < 0.3 and _d < max_deg:
                b.when_first_row().assert_zero(e)
            elif r < 0.4 and _d < max_deg:
                b.when_last_row().assert_zero(e)
            else:
                b.assert_zero(e)

    # lookup expressions may only use the current row of the main and preprocessed traces
    latoms = [E.main(i) for i in range(w)] + [E.var(fe.SRC_PRE, 0, i) for i in range(pw)]
    lookups = []
    for _ in range(int(rng.integers(0, 40 if BIG and rng.random() < 0.3 else 5))):
        m, _ = random_expr(rng, fe, latoms, 1, 1)
        args = [random_expr(rng, fe, latoms, 2, 2)[0] for _ in range(int(rng.integers(0, 70 if BIG and rng.random() < 0.1 else 6)))]
        lookups.append(fe.Lookup.push(m, args) if rng.random() < 0.5 else fe.Lookup.pull(m, args))
    return fe.lookup_air(w, ev, lookups, pre), w, h if pw else None


def one_case(pkg, fe, oracle, ctx, rng, case):
    lb = int(rng.integers(1, 4))
    # FUZZ_ARITY=1: FRI rounds of arity up to 2^6 (drawn from the case number, so the rest of the case is the same system)
    mla = 1 + int(np.random.default_rng(1000 + int(case)).integers(0, 6)) if os.environ.get("FUZZ_ARITY") else 1
    params = fe.Params(log_blowup=lb, cap_height=int(rng.integers(0, 3)), log_final_poly_len=int(rng.choice([0, 0, 0, 1, 2])), max_log_arity=mla,
                       num_queries=int(rng.integers(1, 24)), commit_proof_of_work_bits=int(rng.integers(0, 7)),
                       query_proof_of_work_bits=int(rng.integers(0, 7)))
    if os.environ.get("FUZZ_PARAMS"):  # (drawn from the case number: the rest of the case is the system the plain run draws)
        pr = np.random.default_rng(3000 + int(case))
        params = fe.Params(log_blowup=lb, cap_height=int(pr.integers(0, 7)), log_final_poly_len=int(pr.integers(0, 6)), max_log_arity=mla,
                           num_queries=int(pr.choice([1, 7, 33, 64, 100, 120])), commit_proof_of_work_bits=int(pr.integers(0, 13)),
                           query_proof_of_work_bits=int(pr.integers(0, 13)))
    circuits, traces = [], []
    # FUZZ_MANY=1: systems of 4 .. 40 circuits (many trace heights, many FRI inputs, long matrix lists) instead of 1 .. 3
    n_circuits = int(np.random.default_rng(2000 + int(case)).integers(4, 41)) if os.environ.get("FUZZ_MANY") else int(rng.integers(1, 4))
    for _ in range(n_circuits):
        ci, w, fixed_h = random_circuit(rng, fe, lb)
        circuits.append(ci)
        h = fixed_h if fixed_h else 1 << int(rng.integers(0, 16 if BIG else 11))
        if rng.random() < 0.12:
            h = 0  # inactive circuit
        traces.append(rand_field(rng, (h, w)))
    if all(t.shape[0] == 0 for t in traces):
        traces[0] = rand_field(rng, (circuits[0].preprocessed.shape[0] if circuits[0].preprocessed is not None else 4, traces[0].shape[1]))
    claims = [[int(x) for x in rand_field(rng, int(rng.integers(0, 6)))] for _ in range(int(rng.integers(0, 5)))]
    if LONG_CLAIMS and not BABYBEAR:
        # few long claims or many short ones, 8200 .. 20000 words in all (csrc/outer.hip takes over above 8192)
        n_cl = int(rng.choice([1, 2, 7, 100, 255, 256, 257, 700, 3000]))
        total = int(rng.integers(8200, 20000))
        cuts = np.sort(rng.integers(0, total + 1, n_cl - 1)) if n_cl > 1 else np.array([], dtype=np.int64)
        lens = np.diff(np.concatenate([[0], cuts, [total]]))
        claims = [[int(x) for x in rand_field(rng, int(m))] for m in lens]
    packed = fe.pack_claims(claims)
    try:
        compiled = [fe.compile_circuit(c) for c in circuits]
    except fe.CompileError:
        return "front-end-rejected"  # e.g. a constraint that folded to a non-zero constant (src/graph.rs)
    blob = fe.system_blob(params, compiled, KPERM) if BABYBEAR else fe.system_blob(params, compiled)
    try:
        g = (pkg.babybear.System if BABYBEAR else pkg.System)(ctx, blob, len(compiled))
    except pkg.MstarkError as e:
        o_failed = False
        try:
            oracle.System(blob)
        except Exception:
            o_failed = True
        assert o_failed, "library rejected a system the oracle accepts: %s" % e
        return "rejected-by-both"
    o = oracle.System(g.blob)
    try:
        want = o.prove(traces, packed)
    except RuntimeError as oe:
        # inputs the reference would panic on (e.g. a trace shorter than the final FRI layer): both sides must refuse
        try:
            g.prove_multiple_claims(g.witness(traces, packed))
        except pkg.MstarkError:
            return "prove-refused-by-both"
        raise AssertionError("case %d: the oracle refused (%s) but the library produced a proof" % (case, oe))
    got = g.prove_multiple_claims(g.witness(traces, packed)).to_bytes()
    assert got == want, "case %d: proof bytes differ (len %d vs %d)" % (case, len(got), len(want))
    if os.environ.get("FUZZ_LEVEL2"):
        sys.path.insert(0, os.path.join(ROOT, "tests")) if os.path.join(ROOT, "tests") not in sys.path else None
        if BABYBEAR:
            import test_gpu_bb_level2 as l2

            g.n_circuits = len(compiled)
            step_bytes = l2.level2_prove(g, params, traces, packed)[0]
        else:
            import test_gpu_level2 as l2

            g.params = params  # (System.new sets it; this system came from its blob)
            step_bytes = l2.level2_prove(pkg, ctx, g, params, traces, packed)[0]
        assert step_bytes == want, "case %d: the Level-2 loop's proof differs" % case
    if os.environ.get("FUZZ_FAULTS"):
        fr = np.random.default_rng(4000 + int(case))
        make = (lambda: g.host_witness(traces, packed)) if case % 2 else (lambda: g.witness(traces, packed))
        w_f = make()
        for nth in sorted(int(x) for x in fr.integers(1, 160, 3)):
            ctx.debug_fail_alloc(nth)
            try:
                again = g.prove_multiple_claims(w_f).to_bytes()
                assert again == want, "case %d: proof differs with an unreached injected failure (%d)" % (case, nth)
            except pkg.MstarkError as e:
                assert "injected" in str(e), "case %d: %s" % (case, e)
            finally:
                ctx.debug_fail_alloc(0)
            assert g.prove_multiple_claims(w_f).to_bytes() == want, "case %d: proof differs after an injected failure (%d)" % (case, nth)
    if os.environ.get("FUZZ_PARAMS") and case % 3 == 0:
        assert g.prove_multiple_claims(g.host_witness(traces, packed)).to_bytes() == want, "case %d: host-resident witness: proof differs" % case
    b = o.verify(packed, got)
    a = g.verify(packed, got)  # the product verifier of the configuration (ms_verify / msbb_verify)
    assert (a == 0) == (b == 0), "case %d: verifier verdicts differ: library %d, oracle %d" % (case, a, b)
    return "verified" if a == 0 else "proved"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pkg = load_package()
    fe = pkg.frontend
    global KPERM
    if BABYBEAR:
        import contextlib
        import oracle_bb as oracle
        KPERM = fe.poseidon2_constants()
    else:
        import contextlib
        import oracle

    ctx = pkg.Context(0)
    rng = np.random.default_rng(seed)
    t0 = time.time()
    tally = {}
    only = {int(x) for x in os.environ["FUZZ_ONLY"].split(",")} if os.environ.get("FUZZ_ONLY") else None  # (replay of chosen cases)
    for case in range(n):
        sub = np.random.default_rng(rng.integers(0, 1 << 62))
        if only is not None and case not in only:
            continue
        if only is not None:
            print("[fuzz] case %d" % case, flush=True)
        with (fe.field(fe.BABYBEAR) if BABYBEAR else contextlib.nullcontext()):
            r = one_case(pkg, fe, oracle, ctx, sub, case)
        tally[r] = tally.get(r, 0) + 1
        if (case + 1) % 20 == 0:
            print("[fuzz %6.1fs] %d cases: %s" % (time.time() - t0, case + 1, tally), flush=True)
    print("OK: %d random systems, byte-identical proofs, agreeing verifiers: %s" % (n, tally))


if __name__ == "__main__":
    main()
