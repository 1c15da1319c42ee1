"""GPU parity of the Level-2 entry points (include/mstark.h, "the prover's steps on device handles"): the reference's prover
loop (src/prover.rs:290-603) is driven from Python here - every stage one call, the transcript through ms_challenger_*,
traces / LDEs / trees staying in HBM behind handles - and must yield exactly the bytes ms_prove writes (which the other
tests compare with the oracle). This is the shape of the north star's "Rust host keeps the bookkeeping and calls the
kernels through FFI"."""
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = (1 << 64) - (1 << 32) + 1
W32 = 1753635133440165772  # two_adic_generator(32)


def _gen(bits):
    g = W32
    for _ in range(32 - bits):
        g = g * g % P
    return g


def _round_bytes(vals, widths, npoints):
    """one OpenedValues round from the flat (c0, c1) stream: matrix -> point -> column"""
    out = [struct.pack("<Q", len(widths))]
    for w, np_ in zip(widths, npoints):
        out.append(struct.pack("<Q", np_))
        for _ in range(np_):
            out.append(struct.pack("<Q", w))
            for _ in range(w):
                out.append(struct.pack("<QQ", next(vals), next(vals)))
    return b"".join(out)


def level2_prove(pkg, ctx, system, params, traces, packed):
    n = system.n_circuits
    infos = [system.circuit_info(i) for i in range(n)]
    lb, cap_h = params.log_blowup, params.cap_height
    ch = pkg.Challenger(params)
    ch.observe([n])                                                           # src/system.rs:211-222
    for inf in infos:
        ch.observe([inf[k] for k in ("constraint_count", "max_constraint_degree", "pre_height", "pre_width", "main_width", "stage2_width")])
    active = [t.shape[0] > 0 for t in traces]
    ch.observe([1 if a else 0 for a in active])
    aidx = [i for i in range(n) if active[i]]
    log_degrees = [int(traces[i].shape[0]).bit_length() - 1 for i in aidx]
    w = system.witness(traces, packed)
    # stage 1 (src/prover.rs:336-351)
    s1 = w.commit_stage1([traces[i].shape[0] for i in aidx], [infos[i]["main_width"] for i in aidx])
    pre_cap = system.preprocessed_commit()
    if pre_cap:
        ch.observe_digests(pre_cap)
    ch.observe_digests(s1.cap)
    ch.observe(log_degrees)
    ch.observe_claims(w)                                                      # :369-373
    beta = ch.sample_ext()
    ch.observe(beta)
    gamma = ch.sample_ext()
    ch.observe(gamma)
    acc0 = w.claims_accumulator(beta, gamma)                                  # :382-387
    # lookups + stage 2 (:391-421)
    accs, s2_traces = w.stage2_build(len(aidx), beta, gamma, acc0)
    s2 = pkg.pcs_commit_traces(ctx, s2_traces, lb, cap_h)
    ch.observe_digests(s2.cap)
    for a in accs:
        ch.observe(a)
    alpha = ch.sample_ext()
    # quotient (:437-528)
    q_ldes, acc_in = [], acc0
    for pos, ci in enumerate(aidx):
        q_ldes.append(system.quotient(ci, log_degrees[pos], s1, pos, s2, pos, [*beta, *gamma, *acc_in, *accs[pos]], alpha))
        acc_in = accs[pos]
    qd = pkg.pcs_commit_ldes(ctx, q_ldes, cap_h)
    ch.observe_digests(qd.cap)
    # opening (:538-581)
    zeta = ch.sample_ext()
    zn = [((zeta[0] * _gen(ld)) % P, (zeta[1] * _gen(ld)) % P) for ld in log_degrees]
    rounds = [(s1, [[zeta, z] for z in zn]), (s2, [[zeta, z] for z in zn]), (qd, [[zeta] for _ in zn])]
    pre_circuits = [i for i in range(n) if infos[i]["pre_width"]]
    if pre_cap:
        pre = system.preprocessed_mmcs([(infos[i]["pre_height"] << lb, infos[i]["pre_width"]) for i in pre_circuits])
        rounds.append((pre, [[zeta, zn[aidx.index(i)]] if active[i] else [] for i in pre_circuits]))
    opened, fri = pkg.pcs_open(ctx, params, rounds, ch)
    # Proof::to_bytes, field order of src/prover.rs:213-238
    vals = iter(int(x) for x in opened)
    r_s1 = _round_bytes(vals, [infos[i]["main_width"] for i in aidx], [2] * len(aidx))
    r_s2 = _round_bytes(vals, [infos[i]["stage2_width"] for i in aidx], [2] * len(aidx))
    r_q = _round_bytes(vals, [2 * infos[i]["quotient_degree"] for i in aidx], [1] * len(aidx))
    r_pre = _round_bytes(vals, [infos[i]["pre_width"] for i in pre_circuits], [2 if active[i] else 0 for i in pre_circuits]) if pre_cap else b""
    assert next(vals, None) is None

    def cap(c):
        return struct.pack("<Q", len(c) // 32) + c

    out = [struct.pack("<Q", n), bytes(1 if a else 0 for a in active), cap(s1.cap), cap(s2.cap), cap(qd.cap),
           struct.pack("<Q", len(accs))] + [struct.pack("<QQ", *a) for a in accs]
    out += [struct.pack("<Q", len(log_degrees)), bytes(log_degrees), fri, r_q, bytes([1 if pre_cap else 0]), r_pre, r_s1, r_s2]
    return b"".join(out), w


@pytest.mark.parametrize("case", ["bench12", "bench6_cap", "even_odd_dead", "squares", "pythagorean", "byte_ops"])
def test_level2_loop_yields_the_proof_of_ms_prove(pkg, ctx, oracle, fe, case):
    params = fe.test_params()
    if case == "bench12":
        traces, claims = fe.u32_add_bench_witness(1 << 12)   # claims long enough to be hashed on the device
        inputs, params = fe.u32_add_system_inputs(), fe.bench_params()
    elif case == "bench6_cap":
        traces, claims = fe.u32_add_bench_witness(1 << 6)
        inputs = fe.u32_add_system_inputs()
        params = fe.Params(log_blowup=2, cap_height=2, log_final_poly_len=2, num_queries=20, commit_proof_of_work_bits=3,
                           query_proof_of_work_bits=5)
    elif case == "even_odd_dead":
        traces, claims = fe.even_odd_traces() + [np.zeros((0, 6), dtype=np.uint64)], [[0, 4, 1]]
        inputs = fe.even_odd_inputs(with_dead=True)
    elif case == "squares":
        traces, claims, inputs = fe.squares_traces(64), [], fe.squares_inputs()
    elif case == "byte_ops":
        traces, claims = fe.byte_operations_witness([(0, 10, 5), (1, 30, 20), (2, 100, 40), (3, 200, 100)])
        inputs = fe.byte_operations_inputs()
    else:
        traces, claims, inputs = [fe.pythagorean_trace(64)], [], fe.pythagorean_inputs()
    system = pkg.System.new(ctx, params, inputs)
    packed = fe.pack_claims(claims)
    got, w = level2_prove(pkg, ctx, system, params, traces, packed)
    want = system.prove_multiple_claims(w).to_bytes()
    assert got == want
    assert oracle.System(system.blob).verify(packed, got) == 0


def test_level2_handles_are_checked(pkg, ctx, fe):
    params = fe.bench_params()
    system = pkg.System.new(ctx, params, fe.u32_add_system_inputs())
    traces, claims = fe.u32_add_bench_witness(1 << 6)
    packed = fe.pack_claims(claims)
    w = system.witness(traces, packed)
    accs, s2_traces = w.stage2_build(2, (1, 2), (3, 4), (0, 0))
    assert [t.info() for t in s2_traces] == [(256, 2, 0), (64, 26, 0)]
    s2 = pkg.pcs_commit_traces(ctx, s2_traces, params.log_blowup, 0)
    assert s2_traces[0].info()[0] == 0                      # consumed
    with pytest.raises(pkg.MstarkError):
        pkg.pcs_commit_traces(ctx, s2_traces, params.log_blowup, 0)   # empty handles
    with pytest.raises(pkg.MstarkError):
        pkg.pcs_commit_ldes(ctx, s2_traces, 0)                        # evaluations are not LDEs
    hw = system.host_witness(traces, packed)
    with pytest.raises(pkg.MstarkError, match="device-resident"):
        hw.commit_stage1([256, 64], [1, 14])
    s1 = w.commit_stage1([256, 64], [1, 14])
    with pytest.raises(pkg.MstarkError, match="shape"):
        system.quotient(1, 6, s1, 0, s2, 1, [0] * 8, (1, 0))          # wrong matrix index
    # inputs are validated before anything is shifted, launched or consumed
    P = fe.P
    with pytest.raises(pkg.MstarkError, match="log_n"):
        system.quotient(1, 64, s1, 1, s2, 1, [0] * 8, (1, 0))         # (1 << 64 would be undefined)
    with pytest.raises(pkg.MstarkError, match="non-canonical"):
        system.quotient(1, 6, s1, 1, s2, 1, [0] * 8, (P, 0))          # alpha
    with pytest.raises(pkg.MstarkError, match="non-canonical"):
        w.stage2_build(2, (P, 2), (3, 4), (0, 0))
    with pytest.raises(pkg.MstarkError, match="non-canonical"):
        w.stage2_build(2, (1, 2), (3, 4), (0, P + 5))
    with pytest.raises(pkg.MstarkError, match="non-canonical"):
        w.claims_accumulator((1, 2), (P, 4))
    # a bad handle in the list leaves the good ones unconsumed
    _, fresh = w.stage2_build(2, (1, 2), (3, 4), (0, 0))
    with pytest.raises(pkg.MstarkError):
        pkg.pcs_commit_traces(ctx, [fresh[0], s2_traces[0]], params.log_blowup, 0)   # the second one is empty
    assert fresh[0].info()[0] == 256                                  # still holds its evaluations
    with pytest.raises(pkg.MstarkError, match="twice"):
        pkg.pcs_commit_traces(ctx, [fresh[0], fresh[0]], params.log_blowup, 0)
    assert pkg.pcs_commit_traces(ctx, fresh, params.log_blowup, 0).cap == s2.cap
