"""CPU tests: the reference's identity pins restated against the oracle (same scenario grids, documented PRNG)."""
import numpy as np
import pytest

from conftest import P, rand_field


def _dft_naive(m):
    h, w = m.shape
    g = pow(7, (P - 1) // h, P) if h > 1 else 1
    out = np.zeros_like(m)
    for k in range(h):
        for c in range(w):
            out[k, c] = sum(int(m[j, c]) * pow(g, j * k, P) for j in range(h)) % P
    return out


def test_dft_matches_definition(oracle):
    rng = np.random.default_rng(3)
    for log_h in (0, 1, 2, 3, 5):
        m = rand_field(rng, (1 << log_h, 2))
        assert np.array_equal(oracle.dft_batch(m), _dft_naive(m))
        assert np.array_equal(oracle.dft_batch(oracle.dft_batch(m), inverse=True), m)


# /root/reference/src/prover.rs:975-999: lde_from_coefficients == coset_lde_batch(evals, B, GENERATOR).bit_reverse_rows()
@pytest.mark.parametrize("log_h", [0, 1, 2, 5, 8])
@pytest.mark.parametrize("log_blowup", [1, 2, 3])
@pytest.mark.parametrize("w", [1, 2, 7])
def test_lde_from_coefficients_matches_commit_transform(oracle, log_h, log_blowup, w):
    rng = np.random.default_rng(0)
    h = 1 << log_h
    coeffs = rand_field(rng, (h, w))
    evals = oracle.dft_batch(coeffs)  # coset_dft_batch(coefficients, ONE)
    expected = oracle.coset_lde_bitrev(evals, log_blowup)
    shifted = coeffs.copy()
    for j in range(h):
        s = pow(7, j, P)
        shifted[j] = [(int(x) * s) % P for x in coeffs[j]]
    got = oracle.lde_from_shifted_coefficients(shifted, log_blowup)
    assert np.array_equal(got, expected)
    # and the definition: storage row r holds P(7 * w_N^{bitrev(r)})
    if log_h <= 2 and w == 1:
        N, lN = h << log_blowup, log_h + log_blowup
        wN = pow(7, (P - 1) // N, P)
        for r in range(N):
            k = int(format(r, "0%db" % lN)[::-1], 2) if lN else 0
            x = 7 * pow(wN, k, P) % P
            assert int(got[r, 0]) == sum(int(coeffs[j, 0]) * pow(x, j, P) for j in range(h)) % P


# /root/reference/src/prover.rs:1006-1041: fused gather == coset iDFT -> slice -> scale rows by GENERATOR^r
@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 7])
@pytest.mark.parametrize("q", [1, 2, 4])
@pytest.mark.parametrize("D", [1, 2])
def test_shifted_quotient_slices_matches_naive_composition(oracle, log_n, q, D):
    rng = np.random.default_rng(1)
    n = 1 << log_n
    big = n * q
    evals = rand_field(rng, (big, D))
    coeffs = oracle.dft_batch(evals, inverse=True)  # plain iDFT, then undo the coset shift
    ginv = pow(7, P - 2, P)
    for j in range(big):
        s = pow(ginv, j, P)
        coeffs[j] = [(int(x) * s) % P for x in coeffs[j]]
    expected = np.zeros((n, q * D), dtype=np.uint64)
    for row in range(n):
        s = pow(7, row, P)
        for chunk in range(q):
            expected[row, chunk * D:(chunk + 1) * D] = [(int(x) * s) % P for x in coeffs[chunk * n + row]]
    assert np.array_equal(oracle.shifted_quotient_slices(evals, q), expected)


# /root/reference/src/lookup.rs:697-756: selectors are unnormalised; is_last/(n g) and is_first/n are the Lagrange basis
@pytest.mark.parametrize("log_n", [2, 3, 5, 8])
def test_selector_normalization_constants(oracle, log_n):
    n = 1 << log_n
    g = pow(7, (P - 1) // n, P)
    is_first, is_last, is_trans, inv_van = oracle.selectors_on_coset(log_n, 1)
    N = 2 * n
    wN = pow(7, (P - 1) // N, P)
    ginv = pow(g, P - 2, P)
    for i in (0, 1, N // 2 + 1, N - 1):
        x = 7 * pow(wN, i, P) % P
        zh = (pow(x, n, P) - 1) % P
        assert int(is_first[i]) == zh * pow((x - 1) % P, P - 2, P) % P
        assert int(is_last[i]) == zh * pow((x - ginv) % P, P - 2, P) % P
        assert int(is_trans[i]) == (x - ginv) % P
        assert int(inv_van[i]) == pow(zh, P - 2, P)
        # textbook Lagrange basis at the last row: prod_{k != n-1} (x - g^k) / (g^{n-1} - g^k)
        num = den = 1
        last = pow(g, n - 1, P)
        for k in range(n - 1):
            gk = pow(g, k, P)
            num = num * ((x - gk) % P) % P
            den = den * ((last - gk) % P) % P
        assert int(is_last[i]) * pow(n * g % P, P - 2, P) % P == num * pow(den, P - 2, P) % P


def test_mmcs_mixed_heights_and_injection(oracle):
    """The gen_pcs_refs scenario (/root/reference/src/types.rs:260-281): heights 8/4/2, widths 2/3/1, opened at 5;
    checked against a by-hand restatement of compress-and-inject."""
    m0 = np.zeros((8, 2), dtype=np.uint64)
    m0[5] = [11, 12]
    m1 = np.zeros((4, 3), dtype=np.uint64)
    m1[2] = [107, 108, 109]
    m2 = np.zeros((2, 1), dtype=np.uint64)
    m2[1] = [202]
    t = oracle.Mmcs([m0, m1, m2])
    H, Cc = oracle.hash_elems, oracle.compress2
    l0 = [H(m0[i]) for i in range(8)]
    l1 = [Cc(Cc(l0[2 * i], l0[2 * i + 1]), H(m1[i])) for i in range(4)]
    l2 = [Cc(Cc(l1[2 * i], l1[2 * i + 1]), H(m2[i])) for i in range(2)]
    root = Cc(l2[0], l2[1])
    assert t.cap == root
    vals, proof = t.open(5)
    assert [int(x) for x in vals] == [11, 12, 107, 108, 109, 202]
    assert proof == l0[4] + l1[3] + l2[0]
    assert t.verify(5, vals, proof) == 1
    bad = vals.copy()
    bad[3] += 1
    assert t.verify(5, bad, proof) == 0


def test_direct_logup_matches_schoolbook(oracle, fe):
    """/root/reference/src/lookup.rs:763-867 restated: the oracle's direct logUp evaluation inside quotient_values ==
    evaluating the synthesized chained-accumulator constraints with genuine extension arithmetic, at random points."""
    rng = np.random.default_rng(11)
    inputs = fe.even_odd_inputs()
    comp = [fe.compile_circuit(ci) for ci in inputs]
    o = oracle.System(fe.system_blob(fe.test_params(), comp))
    ci, log_n, log_q = 0, 2, 1
    info = o.circuit_info(ci)
    N = 1 << (log_n + log_q)
    s1 = rand_field(rng, (N, info["main_width"]))
    s2 = rand_field(rng, (N, info["stage2_width"]))
    publics = [int(x) for x in rng.integers(1, P, 8, dtype=np.uint64)]
    alpha = [int(x) for x in rng.integers(1, P, 2, dtype=np.uint64)]
    got = oracle.quotient_values(o, ci, publics, log_n, log_q, None, s1, s2, alpha)

    # schoolbook: evaluate user constraints + synthesized logUp constraints per row, fold with alpha, times 1/Z_H
    def emul(a, b):
        return ((a[0] * b[0] + 7 * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)

    def eadd(a, b):
        return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)

    def esub(a, b):
        return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)

    is_first, is_last, is_trans, inv_van = oracle.selectors_on_coset(log_n, log_q)
    n = 1 << log_n
    g = pow(7, (P - 1) // n, P)
    inj_norm = pow(n * g % P, P - 2, P)
    beta, gamma = (publics[0], publics[1]), (publics[2], publics[3])
    acc_i, acc_f = (publics[4], publics[5]), (publics[6], publics[7])
    delta = esub(acc_f, acc_i)
    cc = comp[ci]

    def eval_node_vals(i, inext):
        vals = []
        for (kind, source, offset, a, b) in cc.nodes:
            row = inext if offset else i
            if kind == fe.N_CONST:
                v = a
            elif kind == fe.N_VAR:
                v = int(s1[row, a]) if source == fe.SRC_MAIN else int(s2[row, a])
            elif kind == fe.N_PUBLIC:
                v = publics[a]
            elif kind == fe.N_IS_FIRST:
                v = int(is_first[i])
            elif kind == fe.N_IS_LAST:
                v = int(is_last[i])
            elif kind == fe.N_IS_TRANS:
                v = int(is_trans[i])
            elif kind == fe.N_ADD:
                v = (vals[a] + vals[b]) % P
            elif kind == fe.N_SUB:
                v = (vals[a] - vals[b]) % P
            elif kind == fe.N_MUL:
                v = vals[a] * vals[b] % P
            else:
                v = (-vals[a]) % P
            vals.append(v)
        return vals

    for i in range(N):
        inext = (i + (1 << log_q)) % N
        nv = eval_node_vals(i, inext)
        cvs = [nv[z] for z in cc.zeros]
        L = len(cc.lookups)
        last_norm = int(is_last[i]) * inj_norm % P
        inj = (delta[0] * last_norm % P, delta[1] * last_norm % P)
        for j, (m, args) in enumerate(cc.lookups):
            src = (int(s2[i, 2 * j]), int(s2[i, 2 * j + 1]))
            if j < L - 1:
                tgt = (int(s2[i, 2 * j + 2]), int(s2[i, 2 * j + 3]))
            else:
                tgt = eadd((int(s2[inext, 0]), int(s2[inext, 1])), inj)
            f = (0, 0)
            for a in reversed(args):
                f = eadd(emul(f, gamma), (nv[a], 0))
            c = esub(emul(eadd(beta, f), esub(tgt, src)), (nv[m], 0))
            cvs += [c[0], c[1]]
        acc = (0, 0)
        for v in cvs:
            acc = eadd(emul(acc, tuple(alpha)), (v, 0))
        iv = int(inv_van[i])
        assert (int(got[i, 0]), int(got[i, 1])) == (acc[0] * iv % P, acc[1] * iv % P)
