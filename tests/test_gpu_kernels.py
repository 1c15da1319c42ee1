"""GPU parity tests, kernel level: every HIP entry point of include/mstark.h against the oracle on the same
seeded inputs (bit-exact: all arithmetic is integer). Run with `pytest -m gpu`."""
import numpy as np
import pytest

from conftest import P, rand_field

pytestmark = pytest.mark.gpu


def test_field_ops(ctx, oracle):
    rng = np.random.default_rng(1)
    a = rand_field(rng, 4096)
    b = rand_field(rng, 4096)
    ai, bi = [int(x) for x in a], [int(x) for x in b]
    assert [int(x) for x in ctx.field_op(0, a, b)] == [(x + y) % P for x, y in zip(ai, bi)]
    assert [int(x) for x in ctx.field_op(1, a, b)] == [(x - y) % P for x, y in zip(ai, bi)]
    assert [int(x) for x in ctx.field_op(2, a, b)] == [(x * y) % P for x, y in zip(ai, bi)]
    nz = np.where(a == 0, np.uint64(5), a)
    assert [int(x) for x in ctx.field_op(3, nz)] == [pow(int(x), P - 2, P) for x in nz]
    # Ext2: (a0 + a1 X)(b0 + b1 X) mod X^2 - 7
    e = ctx.field_op(4, a, b).reshape(-1, 2)
    for k in range(0, 2048, 97):
        a0, a1, b0, b1 = ai[2 * k], ai[2 * k + 1], bi[2 * k], bi[2 * k + 1]
        assert (int(e[k, 0]), int(e[k, 1])) == ((a0 * b0 + 7 * a1 * b1) % P, (a0 * b1 + a1 * b0) % P)
    inv = ctx.field_op(5, nz).reshape(-1, 2)
    prod = ctx.field_op(4, nz, inv.reshape(-1)).reshape(-1, 2)
    assert np.all(prod[:, 0] == 1) and np.all(prod[:, 1] == 0)


@pytest.mark.parametrize("log_h", [0, 1, 2, 5, 8, 11, 12, 13, 16, 20, 21, 22])
@pytest.mark.parametrize("w", [1, 3])
def test_dft_batch(ctx, oracle, log_h, w):
    rng = np.random.default_rng(100 + log_h)
    m = rand_field(rng, (1 << log_h, w))
    got = ctx.dft_batch(m)
    assert np.array_equal(got, oracle.dft_batch(m))
    back = ctx.dft_batch(got, inverse=True)
    assert np.array_equal(back, m)


# heights of 12 + k bits, k = 1 .. 7: the strided pass of k <= 6 bits runs in registers on shifts alone (ntt_small_strided_k: every
# root of order <= 64 is a power of two), k = 7 and MSAMD_NO_NTT_SMALL=1 take the LDS kernel; 2^26 = 12 + 8 + 6 has the same last pass
@pytest.mark.parametrize("log_h", [13, 14, 15, 16, 17, 18, 19, 26])
@pytest.mark.parametrize("small", [True, False])
def test_dft_batch_small_strided_passes(ctx, oracle, log_h, small, monkeypatch):
    if not small:
        monkeypatch.setenv("MSAMD_NO_NTT_SMALL", "1")
    if log_h == 26 and not small:
        pytest.skip("one 2^26 case is enough")
    rng = np.random.default_rng(300 + log_h)
    m = rand_field(rng, (1 << log_h, 1 if log_h == 26 else 2))
    got = ctx.dft_batch(m)
    if log_h <= 19:
        assert np.array_equal(got, oracle.dft_batch(m))
    else:  # (the oracle at 2^26 would take a minute: a size-independent property instead - linearity against a shifted input, and the round trip)
        m2 = m.copy()
        m2[12345, 0] = np.uint64((int(m[12345, 0]) + 1) % P)
        g2 = ctx.dft_batch(m2)
        w = int(oracle.lib().mso_gl_two_adic_generator(log_h))
        for k in (0, 1, 2, 77, (1 << 25) + 3, (1 << 26) - 1):
            assert (int(g2[k, 0]) - int(got[k, 0])) % P == pow(w, 12345 * k, P)
    assert np.array_equal(ctx.dft_batch(got, inverse=True), m)
    if log_h <= 17:
        for lb in (1, 2):
            assert np.array_equal(ctx.coset_lde_batch(m, lb), oracle.coset_lde_bitrev(m, lb))


# the reference's layout pin: src/prover.rs:975-999 (h in {1,2,4,32,256}, B in {2,4,8}, w in {1,2,7})
@pytest.mark.parametrize("log_h", [0, 1, 2, 5, 8, 12, 14])
@pytest.mark.parametrize("log_blowup", [1, 2, 3])
@pytest.mark.parametrize("w", [1, 2, 7])
def test_coset_lde_matches_commit_transform(ctx, oracle, log_h, log_blowup, w):
    rng = np.random.default_rng(1000 * log_h + 10 * log_blowup + w)
    m = rand_field(rng, (1 << log_h, w))
    assert np.array_equal(ctx.coset_lde_batch(m, log_blowup), oracle.coset_lde_bitrev(m, log_blowup))


# src/prover.rs:1006-1041 scenario grid (n in {1,2,4,32,128}, q in {1,2,4}, D in {1,2}) + larger sizes
@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 7, 12])
@pytest.mark.parametrize("log_q", [0, 1, 2])
@pytest.mark.parametrize("D", [1, 2])
def test_quotient_lde(ctx, oracle, log_n, log_q, D):
    rng = np.random.default_rng(7 + log_n * 31 + log_q * 5 + D)
    q = rand_field(rng, (1 << (log_n + log_q), D))
    for lb in (1, 2):
        if lb < log_q:
            continue
        want = oracle.lde_from_shifted_coefficients(oracle.shifted_quotient_slices(q, 1 << log_q), lb)
        assert np.array_equal(ctx.quotient_lde(q, log_n, log_q, lb), want)


@pytest.mark.parametrize("n", [0, 1, 3, 63, 64, 65, 1023, 1024, 1025, 2048, 2049, 3072, 5000, 7 * 1024 + 1, 100003])
def test_blake3_stream(ctx, oracle, n):
    data = bytes((i * 7 + 3) % 251 for i in range(n))
    assert ctx.blake3(data) == oracle.hash_bytes(data)


@pytest.mark.parametrize("shapes", [
    [(8, 2), (4, 3), (2, 1)],              # the reference's gen_pcs_refs scenario (src/types.rs:260-275)
    [(1, 1)],
    [(64, 5)],
    [(1024, 14), (1024, 2), (16, 1)],
    [(256, 130)],                          # rows longer than one BLAKE3 chunk
    [(64, 300), (64, 1), (8, 129)],
    [(4096, 26), (1024, 2)],
    # the one-launch sub-tree kernel (hash.hip::subtree_k): single workgroup, several workgroups with the in-launch
    # hand-over of their roots, an injected group in the first level, inside a sub-tree, among the roots and at the root;
    # two injected groups take the layer-by-layer path
    [(2, 1)],
    [(2048, 1)],
    [(8192, 3), (2048, 2)],
    [(4096, 2), (2048, 1)],
    [(16384, 2), (4, 130)],
    [(4096, 1), (1, 3)],
    [(1 << 15, 1), (1 << 13, 2), (4, 1)],
    [(1 << 17, 2), (1 << 9, 5)],
    # few rows of many BLAKE3 chunks each, injected: chunk-parallel row digests in front of the tree launch (wide_rows_*_k) -
    # one matrix, a group of three whose chunks straddle the matrices, a last chunk of one element, the first level
    [(1 << 12, 3), (1 << 8, 700)],
    [(1 << 13, 2), (1 << 7, 300), (1 << 7, 129), (1 << 7, 212)],
    [(1 << 11, 1), (1 << 10, 513)],
    [(1 << 14, 1), (4, 2625)],
])
@pytest.mark.parametrize("cap_height", [0, 2])
def test_mmcs_commit_open(pkg, ctx, oracle, shapes, cap_height):
    rng = np.random.default_rng(len(shapes) * 17 + cap_height)
    mats = [rand_field(rng, s) for s in shapes]
    g = pkg.Mmcs(ctx, mats, cap_height)
    o = oracle.Mmcs(mats, cap_height)
    assert g.cap == o.cap
    maxh = max(s[0] for s in shapes)
    for index in sorted({0, 1 % maxh, maxh // 2, maxh - 1, 5 % maxh}):
        gv, gp = g.open(index)
        ov, op = o.open(index)
        assert np.array_equal(gv, ov) and gp == op
        if min(s[0] for s in shapes) >= (1 << cap_height):   # shorter matrices are not bound by a cap this tall
            assert o.verify(index, gv, gp, g.cap) == 1


@pytest.mark.parametrize("h,L,widths", [(1, 1, [2]), (4, 2, [3, 3]), (256, 1, [2]), (64, 13, [4] + [2] * 12),
                                        (4096, 13, [4] + [2] * 12), (32, 3, [0, 5, 1]), (8192, 9, [1] * 9)])
def test_stage2_trace(ctx, oracle, h, L, widths):
    rng = np.random.default_rng(h + L)
    offs = np.concatenate([[0], np.cumsum(widths)]).astype(np.uint64)
    mult = rand_field(rng, (h, L))
    args = rand_field(rng, (h, max(int(offs[-1]), 1)))[:, : int(offs[-1])]
    # challenges are uniformly random in the protocol: no edge values here (a zero message has no inverse)
    beta, gamma, acc = [int(x) for x in rng.integers(2, P, 2, dtype=np.uint64)], [int(x) for x in rng.integers(2, P, 2, dtype=np.uint64)], [3, 9]
    gt, ga = ctx.stage2_trace(mult, offs, args, beta, gamma, acc)
    ot, oa = oracle.stage2_trace(mult, offs, args, beta, gamma, acc)
    assert ga == oa
    assert np.array_equal(gt, ot)


def test_stage2_no_lookups(ctx, oracle):
    mult = np.zeros((16, 0), dtype=np.uint64)
    gt, ga = ctx.stage2_trace(mult, np.zeros(1, dtype=np.uint64), np.zeros((16, 0), dtype=np.uint64), [1, 2], [3, 4], [5, 6])
    assert ga == (5, 6) and gt.shape == (16, 2) and not gt.any()


@pytest.mark.parametrize("n", [0, 1, 7, 256, 257, 5000, 70000])
def test_claims_accumulator(ctx, oracle, fe, n):
    rng = np.random.default_rng(n)
    claims = [list(map(int, rand_field(rng, int(rng.integers(0, 6))))) for _ in range(n)]
    packed = fe.pack_claims(claims)
    beta, gamma = [11, 12], [13, 14]
    assert ctx.claims_accumulator(packed, beta, gamma) == oracle.claims_accumulator(packed, beta, gamma)


def _systems(pkg, ctx, oracle, fe, inputs, params):
    g = pkg.System.new(ctx, params, inputs)
    return g, oracle.System(g.blob)


@pytest.mark.parametrize("which,log_n", [("pyth", 2), ("pyth", 6), ("evenodd", 2), ("evenodd", 5), ("u32", 3), ("u32", 8), ("u32", 12),
                                         ("u32", 20)])
def test_quotient_values(pkg, ctx, oracle, fe, which, log_n):
    rng = np.random.default_rng(log_n)
    inputs = {"pyth": fe.pythagorean_inputs, "evenodd": fe.even_odd_inputs, "u32": fe.u32_add_system_inputs}[which]()
    params = fe.bench_params()
    g, o = _systems(pkg, ctx, oracle, fe, inputs, params)
    for ci in range(len(inputs)):
        info = g.circuit_info(ci)
        assert info == o.circuit_info(ci)
        ln = 8 if info["pre_height"] else log_n   # byte table is fixed at 256 rows
        lq = info["quotient_degree"].bit_length() - 1
        N = 1 << (ln + lq)
        pre = rand_field(rng, (N, info["pre_width"])) if info["pre_width"] else None
        s1 = rand_field(rng, (N, info["main_width"]))
        s2 = rand_field(rng, (N, info["stage2_width"]))
        publics = rand_field(rng, 8)
        alpha = rand_field(rng, 2)
        want = oracle.quotient_values(o, ci, publics, ln, lq, pre, s1, s2, alpha)
        got = g.quotient_values(ci, publics, ln, lq, pre, s1, s2, alpha)
        assert np.array_equal(got, want)


# ---- the same entry points at the bench's full sizes (config 2: 2^20-row traces, 2^22-row LDEs), still bit-exact against
# the oracle: these are the launches bench.py times
def test_full_size_coset_lde(ctx, oracle):
    rng = np.random.default_rng(2020)
    m = rand_field(rng, (1 << 20, 2))
    assert np.array_equal(ctx.coset_lde_batch(m, 2), oracle.coset_lde_bitrev(m, 2))


def test_full_size_merkle_tree(pkg, ctx, oracle):
    rng = np.random.default_rng(2022)
    mats = [rng.integers(0, P, (1 << 22, 14), dtype=np.uint64), rng.integers(0, P, (1024, 1), dtype=np.uint64)]
    g = pkg.Mmcs(ctx, mats, 0)
    o = oracle.Mmcs(mats, 0)
    assert g.cap == o.cap
    for index in (0, 1, (1 << 22) - 1, 1234567, 3 << 20):
        gv, gp = g.open(index)
        ov, op = o.open(index)
        assert np.array_equal(gv, ov) and gp == op and len(gp) == 22 * 32


def test_full_size_stage2_trace(ctx, oracle):
    rng = np.random.default_rng(2021)
    h, widths = 1 << 20, [4] + [2] * 12
    offs = np.concatenate([[0], np.cumsum(widths)]).astype(np.uint64)
    mult = rng.integers(0, 2, (h, 13), dtype=np.uint64)
    args = rng.integers(0, 1 << 32, (h, int(offs[-1])), dtype=np.uint64)
    beta, gamma = [int(x) for x in rng.integers(2, P, 2, dtype=np.uint64)], [int(x) for x in rng.integers(2, P, 2, dtype=np.uint64)]
    gt, ga = ctx.stage2_trace(mult, offs, args, beta, gamma, [3, 9])
    ot, oa = oracle.stage2_trace(mult, offs, args, beta, gamma, [3, 9])
    assert ga == oa and np.array_equal(gt, ot)


# ---- Pcs::commit / open / verify on their own (ms_pcs_*, ms_challenger_*)
from __graft_entry__ import load_package as _load_package  # noqa: E402

_fe = _load_package().frontend

def _pcs_scenario(ctx, pkg, oracle, params, mats_per_round, n_points):
    """commit every round, observe the commitments, sample zeta, open: library vs oracle, then both verifiers on both results"""
    lb, ch_h = params.log_blowup, params.cap_height
    g_rounds = [pkg.PcsCommitment(ctx, mats, lb, ch_h) for mats in mats_per_round]
    o_rounds = [oracle.Mmcs([oracle.coset_lde_bitrev(m, lb) for m in mats], ch_h) for mats in mats_per_round]
    for g, o in zip(g_rounds, o_rounds):
        assert g.cap == o.cap
    gc, oc = pkg.Challenger(params), oracle.Challenger.for_params(params)
    for g in g_rounds:
        gc.observe_digests(g.cap)
        oc.observe_digests(g.cap)
    zeta = gc.sample_ext()
    assert zeta == oc.sample_ext()
    pts_per_round = []
    for mats in mats_per_round:
        per = []
        for m in mats:
            log_n = m.shape[0].bit_length() - 1
            g_n = pow(oracle.lib().mso_gl_two_adic_generator(log_n), 1, P)
            zn = (zeta[0] * g_n % P, zeta[1] * g_n % P)
            per.append([zeta, zeta][:n_points] if n_points <= 2 and n_points != -1 else [zeta, zn])
        pts_per_round.append(per)
    g_open, g_fri = pkg.pcs_open(ctx, params, list(zip(g_rounds, pts_per_round)), gc)
    o_open, o_fri = oracle.pcs_open(params, list(zip(o_rounds, pts_per_round)), oc)
    assert np.array_equal(g_open, o_open) and g_fri == o_fri
    assert gc.sample_bits(20) == oc.sample_bits(20)  # the transcripts end in the same state
    desc = [(g.cap, [(m.shape[0].bit_length() - 1, m.shape[1]) for m in mats], per)
            for g, mats, per in zip(g_rounds, mats_per_round, pts_per_round)]

    def fresh(lib_side):
        c = pkg.Challenger(params) if lib_side else oracle.Challenger.for_params(params)
        for g in g_rounds:
            c.observe_digests(g.cap)
        assert c.sample_ext() == zeta
        return c

    assert pkg.pcs_verify(params, desc, g_open, g_fri, fresh(True)) and oracle.pcs_verify(params, desc, g_open, g_fri, fresh(False))
    bad = g_open.copy()
    bad[0] ^= 1
    assert not pkg.pcs_verify(params, desc, bad, g_fri, fresh(True)) and not oracle.pcs_verify(params, desc, bad, g_fri, fresh(False))
    tam = bytearray(g_fri)
    tam[len(tam) // 3] ^= 2
    assert not pkg.pcs_verify(params, desc, g_open, bytes(tam), fresh(True))
    assert not oracle.pcs_verify(params, desc, g_open, bytes(tam), fresh(False))
    assert not pkg.pcs_verify(params, desc, g_open, g_fri[:-3], fresh(True))


def test_pcs_example(ctx, pkg, oracle):
    """examples/pcs_example.rs: one polynomial of degree < 32 (entry i = i), blowup 2, 100 queries, opened twice at zeta"""
    params = _fe.Params(1, 0, 0, 1, 100, 0, 0)
    _pcs_scenario(ctx, pkg, oracle, params, [[np.arange(32, dtype=np.uint64).reshape(32, 1)]], 2)


@pytest.mark.parametrize("params", [_fe.Params(2, 1, 1, 1, 15, 4, 5), _fe.Params(1, 0, 0, 1, 30, 0, 0), _fe.Params(3, 2, 2, 1, 9, 6, 0)])
def test_pcs_commit_open_verify_mixed_rounds(ctx, pkg, oracle, params):
    """two committed batches, mixed heights and widths, one point or (zeta, zeta g) per matrix"""
    rng = np.random.default_rng(int(params.num_queries))
    rounds = [[rand_field(rng, (1 << 10, 5)), rand_field(rng, (1 << 7, 2)), rand_field(rng, (1 << 10, 1))], [rand_field(rng, (1 << 9, 9))]]
    _pcs_scenario(ctx, pkg, oracle, params, rounds, -1)
    _pcs_scenario(ctx, pkg, oracle, params, rounds, 1)


def test_pcs_open_error_paths(ctx, pkg):
    params = _fe.Params(1, 0, 0, 1, 10, 0, 0)
    c = pkg.PcsCommitment(ctx, [np.arange(64, dtype=np.uint64).reshape(32, 2)], 1, 0)
    ch = pkg.Challenger(params)
    ch.observe_digests(c.cap)
    z = ch.sample_ext()
    with pytest.raises(pkg.MstarkError):  # at most two opening points per matrix
        pkg.pcs_open(ctx, params, [(c, [[z, z, z]])], ch)
    with pytest.raises(pkg.MstarkError):  # non-canonical point
        pkg.pcs_open(ctx, params, [(c, [[(P, 0)]])], pkg.Challenger(params))
    with pytest.raises(pkg.MstarkError):  # unsupported folding arity
        pkg.pcs_open(ctx, _fe.Params(1, 0, 0, 7, 10, 0, 0), [(c, [[z]])], pkg.Challenger(params))
    with pytest.raises(pkg.MstarkError):
        ch.observe([P])
    # a matrix that is not taller than blowup * final polynomial length is refused (p3 prove_fri precondition)
    with pytest.raises(pkg.MstarkError):
        pkg.pcs_open(ctx, _fe.Params(1, 0, 5, 1, 10, 0, 0), [(c, [[z]])], pkg.Challenger(params))
