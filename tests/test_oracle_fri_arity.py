"""CPU tests: FRI folding arities above 2 (`FriParameters::max_log_arity`, /root/reference/src/types.rs:189-190,215) in the oracle.

The reference sets `max_log_arity: 1` at every call site and holds no vector for a wider fold, so the row layout, the arity
schedule and the roll-in factor are restated from the published p3-fri algorithm (PARITY UNPINNED). What
CAN be checked here without the reference is checked independently of the oracle's own fold code: a pure-Python replay of
the commit phase that folds every row by LAGRANGE INTERPOLATION at beta (no binary steps) must reproduce every sibling value
and the final polynomial of the oracle's proof."""
import numpy as np
import pytest

import proof_codec
from proof_codec import _R

P = 0xFFFFFFFF00000001
W = 7  # BinomialExtensionField<Goldilocks, 2>: X^2 = 7


def e_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def e_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def e_mul(a, b):
    return ((a[0] * b[0] + W * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def e_inv(a):
    n = pow((a[0] * a[0] - W * a[1] * a[1]) % P, P - 2, P)
    return (a[0] * n % P, (-a[1]) * n % P)


def e_pow(a, k):
    r = (1, 0)
    while k:
        if k & 1:
            r = e_mul(r, a)
        a = e_mul(a, a)
        k >>= 1
    return r


def bitrev(i, bits):
    return int(format(i, "0%db" % bits)[::-1], 2) if bits else 0


def parse_fri(b):
    proof_codec._ELEM, proof_codec._DEG = 8, 2  # the codec's reader takes the field from module state (set by parse())
    r = _R(b)
    f = {"commits": [r.cap() for _ in range(r.cnt())], "pow": [r.fe() for _ in range(r.cnt())], "queries": []}
    for _ in range(r.cnt()):
        q = {"input": [], "steps": []}
        for _ in range(r.cnt()):
            rows = [[r.fe() for _ in range(r.cnt())] for _ in range(r.cnt())]
            q["input"].append((rows, [r.raw(32) for _ in range(r.cnt())]))
        for _ in range(r.cnt()):
            la = r.u8()
            sib = [tuple(r.ext()) for _ in range(r.cnt())]
            q["steps"].append((la, sib, [r.raw(32) for _ in range(r.cnt())]))
        f["queries"].append(q)
    f["final_poly"] = [tuple(r.ext()) for _ in range(r.cnt())]
    f["query_pow"] = r.fe()
    assert r.o == len(b)
    return f


def schedule(log_heights, log_final, max_log_arity):
    """arity of every round for reduced-opening vectors of these log heights (tallest first)"""
    hs = sorted(set(log_heights), reverse=True)
    h, nxt, out = hs[0], 1, []
    while h > log_final:
        a = min(max_log_arity, h - log_final)
        if nxt < len(hs):
            a = min(a, h - hs[nxt])
        out.append(a)
        h -= a
        if nxt < len(hs) and hs[nxt] == h:
            nxt += 1
    return out


@pytest.mark.parametrize("log_n,lb,lfp,mla,commit_pow", [(6, 1, 0, 2, 0), (7, 2, 0, 3, 3), (6, 1, 1, 2, 2), (8, 1, 2, 4, 0), (5, 2, 0, 6, 0)])
def test_fold_is_interpolation_at_beta(oracle, fe, log_n, lb, lfp, mla, commit_pow):
    params = fe.Params(lb, 0, lfp, mla, 12, commit_pow, 0)
    rng = np.random.default_rng(log_n * 100 + mla)
    mat = rng.integers(0, P, size=(1 << log_n, 1), dtype=np.uint64)
    lde = oracle.coset_lde_bitrev(mat, lb)
    mm = oracle.Mmcs([lde], 0)

    def start():
        c = oracle.Challenger.for_params(params)
        c.observe_digests(mm.cap)
        return c, tuple(int(x) for x in c.sample_ext())

    ch, zeta = start()
    opened, fri_bytes = oracle.pcs_open(params, [(mm, [[zeta]])], ch)
    y = (int(opened[0]), int(opened[1]))
    fri = parse_fri(fri_bytes)

    # ---- the one reduced-opening vector, from its definition: (y - f(x_i)) / (zeta - x_i), x_i = 7 w^bitrev(i), alpha^0 = 1
    L = log_n + lb
    w = int(oracle.lib().mso_gl_two_adic_generator(L))
    vec = []
    for i in range(1 << L):
        x = 7 * pow(w, bitrev(i, L), P) % P
        vec.append(e_mul(e_sub(y, (int(lde[i, 0]), 0)), e_inv(e_sub(zeta, (x, 0)))))

    # ---- transcript replay on a second challenger
    ch2, z2 = start()
    assert z2 == zeta
    for v in y:
        ch2.observe(v)
    ch2.sample_ext()  # alpha: multiplies nothing in a one-column, one-point opening
    arities = schedule([L], lb + lfp, mla)
    assert [s[0] for s in fri["queries"][0]["steps"]] == arities and len(fri["commits"]) == len(arities)
    assert max(arities) > 1
    layers = [vec]
    for rnd, a in enumerate(arities):
        ch2.observe_digests(b"".join(fri["commits"][rnd]))
        if commit_pow:
            ch2.observe(fri["pow"][rnd])
            assert ch2.sample_bits(commit_pow) == 0
        beta = tuple(int(x) for x in ch2.sample_ext())
        cur, ar, Lc = layers[-1], 1 << a, L - sum(arities[:rnd])
        wl = int(oracle.lib().mso_gl_two_adic_generator(Lc))
        nxt = []
        for r in range(len(cur) >> a):
            xs = [pow(wl, bitrev(r * ar + j, Lc), P) for j in range(ar)]  # FRI folds over the subgroup (no coset shift)
            acc = (0, 0)
            for j in range(ar):
                num, den = (1, 0), 1
                for k in range(ar):
                    if k != j:
                        num = e_mul(num, e_sub(beta, (xs[k], 0)))
                        den = den * (xs[j] - xs[k]) % P
                acc = e_add(acc, e_mul(e_mul(cur[r * ar + j], num), (pow(den, P - 2, P), 0)))
            nxt.append(acc)
        layers.append(nxt)

    # ---- final polynomial: the last layer, truncated to the final length, is its evaluation in bit-reversed order
    fl = 1 << lfp
    wf = int(oracle.lib().mso_gl_two_adic_generator(lfp))
    for i in range(fl):
        x, acc = pow(wf, bitrev(i, lfp), P), (0, 0)
        for c in reversed(fri["final_poly"]):
            acc = e_add(e_mul(acc, (x, 0)), c)
        assert acc == layers[-1][i]
    # the rest of the last layer is the same polynomial on the other cosets of the final domain
    wF = int(oracle.lib().mso_gl_two_adic_generator(lfp + lb))
    for i in range(len(layers[-1])):
        x, acc = pow(wF, bitrev(i, lfp + lb), P), (0, 0)
        for c in reversed(fri["final_poly"]):
            acc = e_add(e_mul(acc, (x, 0)), c)
        assert acc == layers[-1][i]

    # ---- every query: the opened row of every round is that round's layer (own value left out, row order kept)
    for c in fri["final_poly"]:
        for v in c:
            ch2.observe(v)
    for q in fri["queries"]:
        idx = ch2.sample_bits(L)
        for rnd, (la, sib, path) in enumerate(q["steps"]):
            ar = 1 << la
            row, own = idx >> la, idx & (ar - 1)
            want = [layers[rnd][row * ar + j] for j in range(ar) if j != own]
            assert sib == want, "round %d" % rnd
            assert len(path) == (L - sum(arities[:rnd + 1]))
            idx = row


ARITY_PARAMS = [dict(log_blowup=1, max_log_arity=2), dict(log_blowup=2, max_log_arity=3, commit_proof_of_work_bits=3, query_proof_of_work_bits=2),
                dict(log_blowup=1, max_log_arity=4, log_final_poly_len=1), dict(log_blowup=2, cap_height=2, max_log_arity=2),
                dict(log_blowup=1, max_log_arity=16)]


@pytest.mark.parametrize("kw", ARITY_PARAMS)
def test_wide_folds_prove_and_verify(oracle, fe, kw):
    """whole proofs: mixed trace heights (inputs rolled in between rounds bound the arity), lookups, tampering"""
    params = fe.Params(num_queries=20, **kw)
    comp = [fe.compile_circuit(ci) for ci in fe.even_odd_inputs()]
    s = oracle.System(fe.system_blob(params, comp))
    packed = fe.pack_claims([[0, 4, 1]])
    proof = s.prove(fe.even_odd_traces(), packed)
    assert s.verify(packed, proof) == 0
    assert s.prove(fe.even_odd_traces(), packed) == proof
    rng = np.random.default_rng(11)
    rejected = 0
    for pos in [int(x) for x in rng.integers(0, len(proof), 60)]:
        bad = bytearray(proof)
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        rejected += s.verify(packed, bytes(bad)) != 0
    assert rejected >= 56  # (a proof-of-work witness is not read at zero bits: a flip there is accepted)
    # a verifier configured for another maximum arity refuses the proof (the schedule is part of the statement)
    other = fe.Params(num_queries=20, **dict(kw, max_log_arity=1))
    assert oracle.System(fe.system_blob(other, comp)).verify(packed, proof) != 0

    # taller traces: the bench circuit, where several rounds fold at the full arity
    comp = [fe.compile_circuit(ci) for ci in fe.u32_add_system_inputs()]
    s = oracle.System(fe.system_blob(params, comp))
    traces, claims = fe.u32_add_bench_witness(1 << 7)
    packed = fe.pack_claims(claims)
    proof = s.prove(traces, packed)
    assert s.verify(packed, proof) == 0
    from proof_codec import parse

    pr = parse(proof)
    steps = [st["log_arity"] for st in pr["opening_proof"]["query_proofs"][0]["commit_phase_openings"]]
    # the U32Add trace and the byte table differ in height: the roll-in of the shorter one bounds a round's arity
    assert steps == schedule([ld + params.log_blowup for ld in pr["log_degrees"]], params.log_blowup + params.log_final_poly_len, params.max_log_arity)
    assert max(steps) > 1
    for q in pr["opening_proof"]["query_proofs"]:
        assert [st["log_arity"] for st in q["commit_phase_openings"]] == steps
        assert all(len(st["sibling_values"]) == (1 << st["log_arity"]) - 1 for st in q["commit_phase_openings"])


def test_max_log_arity_range(oracle, fe):
    comp = [fe.compile_circuit(ci) for ci in fe.pythagorean_inputs()]
    for bad in (0, 17):
        with pytest.raises(RuntimeError):
            oracle.System(fe.system_blob(fe.Params(max_log_arity=bad), comp))
