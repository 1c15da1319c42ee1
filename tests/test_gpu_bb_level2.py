"""GPU parity of the Level-2 entry points of the BabyBear / Poseidon2 configuration (include/mstark_bb.h, "the prover's steps on
device handles"): the reference's prover loop (src/prover.rs:290-603) is driven from Python - every stage one call, the
transcript through msbb_challenger_*, traces / LDEs / trees staying in HBM behind handles - and must yield exactly the bytes
msbb_prove writes (which tests/test_gpu_babybear.py compares with the oracle). The counterpart of tests/test_gpu_level2.py."""
import struct

import numpy as np
import pytest

import oracle_bb as ob
from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu
pkg = load_package()
fe = pkg.frontend
bb = pkg.babybear
P = fe.BABYBEAR["P"]
K = fe.poseidon2_constants()
R = (1 << 32) % P  # Montgomery constant: the proof bytes hold x * 2^32 mod p (MontyField31's serde form)


@pytest.fixture(scope="module")
def ctx():
    c = pkg.Context(0)
    bb.set_poseidon2(c, K)
    ob.set_poseidon2(K)
    return c


def _gen(bits):
    """BabyBear::two_adic_generator(bits): 0x1a427a41 has order 2^27"""
    g = 0x1A427A41
    for _ in range(27 - bits):
        g = g * g % P
    return g


def _fe(x):
    return struct.pack("<I", int(x) * R % P)


def _ext(e):
    return b"".join(_fe(x) for x in e)


def _cap(words):
    return struct.pack("<Q", len(words) // 8) + b"".join(_fe(x) for x in words)


def _round_bytes(vals, widths, npoints):
    out = [struct.pack("<Q", len(widths))]
    for w, np_ in zip(widths, npoints):
        out.append(struct.pack("<Q", np_))
        for _ in range(np_):
            out.append(struct.pack("<Q", w))
            for _ in range(w):
                out.append(_ext([next(vals) for _ in range(4)]))
    return b"".join(out)


def level2_prove(system, params, traces, packed):
    n = system.n_circuits
    infos = [system.circuit_info(i) for i in range(n)]
    lb = params.log_blowup
    ch = bb.Challenger(system)
    ch.observe([n])                                                           # src/system.rs:211-222
    for inf in infos:
        ch.observe([inf[k] for k in ("constraint_count", "max_constraint_degree", "pre_height", "pre_width", "main_width", "stage2_width")])
    active = [t.shape[0] > 0 for t in traces]
    ch.observe([1 if a else 0 for a in active])
    aidx = [i for i in range(n) if active[i]]
    log_degrees = [int(traces[i].shape[0]).bit_length() - 1 for i in aidx]
    w = system.witness(traces, packed)
    s1 = bb.commit_stage1(w, params, [traces[i].shape[0] << lb for i in aidx])  # src/prover.rs:336-351
    pre_cap = system.preprocessed_commit()
    if pre_cap is not None:
        ch.observe_digests(pre_cap)
    ch.observe_digests(s1.cap)
    ch.observe(log_degrees)
    ch.observe_claims(w)                                                      # :369-373
    beta = ch.sample_ext()
    ch.observe(beta)
    gamma = ch.sample_ext()
    ch.observe(gamma)
    acc0 = bb.claims_accumulator(w, beta, gamma)                              # :382-387
    accs, s2_traces = bb.stage2_build(w, len(aidx), beta, gamma, acc0)        # :391-409
    s2 = bb.pcs_commit_traces(system, params, s2_traces)                      # :413-421
    ch.observe_digests(s2.cap)
    for a in accs:
        ch.observe(a)
    alpha = ch.sample_ext()
    q_ldes, acc_in = [], acc0                                                 # :437-528
    for pos, ci in enumerate(aidx):
        q_ldes.append(bb.quotient(system, ci, log_degrees[pos], s1, pos, s2, pos, [*beta, *gamma, *acc_in, *accs[pos]], alpha))
        acc_in = accs[pos]
    qd = bb.pcs_commit_ldes(system, params, q_ldes)
    ch.observe_digests(qd.cap)
    zeta = ch.sample_ext()                                                    # :538-581
    zn = [tuple(z * _gen(ld) % P for z in zeta) for ld in log_degrees]
    w_main = [infos[i]["main_width"] for i in aidx]
    w_s2 = [infos[i]["stage2_width"] for i in aidx]
    w_q = [4 * infos[i]["quotient_degree"] for i in aidx]
    rounds = [(s1, w_main, [[zeta, z] for z in zn]), (s2, w_s2, [[zeta, z] for z in zn]), (qd, w_q, [[zeta] for _ in zn])]
    pre_circuits = [i for i in range(n) if infos[i]["pre_width"]]
    if pre_cap is not None:
        pre = bb.preprocessed_mmcs(system)
        rounds.append((pre, [infos[i]["pre_width"] for i in pre_circuits], [[zeta, zn[aidx.index(i)]] if active[i] else [] for i in pre_circuits]))
    opened, fri = bb.pcs_open(system, rounds, ch)
    vals = iter(int(x) for x in opened)
    r_s1 = _round_bytes(vals, w_main, [2] * len(aidx))
    r_s2 = _round_bytes(vals, w_s2, [2] * len(aidx))
    r_q = _round_bytes(vals, w_q, [1] * len(aidx))
    r_pre = _round_bytes(vals, [infos[i]["pre_width"] for i in pre_circuits], [2 if active[i] else 0 for i in pre_circuits]) if pre_cap is not None else b""
    assert next(vals, None) is None
    # Proof::to_bytes, field order of src/prover.rs:213-238
    out = [struct.pack("<Q", n), bytes(1 if a else 0 for a in active), _cap(s1.cap), _cap(s2.cap), _cap(qd.cap),
           struct.pack("<Q", len(accs))] + [_ext(a) for a in accs]
    out += [struct.pack("<Q", len(log_degrees)), bytes(log_degrees), fri, r_q, bytes([1 if pre_cap is not None else 0]), r_pre, r_s1, r_s2]
    return b"".join(out), w, (s2_traces, q_ldes)


@pytest.mark.parametrize("case", ["smoke", "mul_pow", "even_odd", "bench12", "squares_cap"])
def test_level2_loop_yields_the_proof_of_msbb_prove(ctx, case):
    with fe.field(fe.BABYBEAR):
        params = fe.test_params()
        if case == "smoke":            # baby_bear_config.rs:159-206
            inputs, traces, claims = fe.mul_air_inputs(), [fe.mul_air_smoke_trace()], []
        elif case == "mul_pow":        # proof of work on both sides, a 4-coefficient final polynomial, caps
            inputs, traces, claims = fe.mul_air_inputs(), [fe.mul_air_trace(1 << 9)], []
            params = fe.Params(2, 1, 2, 1, 12, 3, 5)
        elif case == "even_odd":       # lookups between circuits, one claim (src/lookup.rs:868-1007)
            inputs, traces, claims = fe.even_odd_inputs(), fe.even_odd_traces(), [[0, 4, 1]]
        elif case == "bench12":        # preprocessed byte table + U32Add, 4096 claims (benches/multi_stark.rs:73-165)
            inputs = fe.u32_add_system_inputs()
            traces, claims = fe.u32_add_bench_witness(1 << 12)
            params = fe.Params(2, 0, 0, 1, 20, 2, 2)
        else:                          # constraint degree 3 -> two quotient slices
            inputs, traces, claims = fe.squares_inputs(), fe.squares_traces(16), []
            params = fe.Params(2, 1, 1, 1, 10, 0, 0)
        system = bb.System.new(ctx, params, inputs, K)
        packed = fe.pack_claims(claims)
    got, w, _ = level2_prove(system, params, traces, packed)
    want = system.prove_multiple_claims(w).to_bytes()
    assert len(got) == len(want)
    assert got == want, "bytes differ at %d" % next(i for i in range(len(got)) if got[i] != want[i])
    assert ob.System(system.blob).verify(packed, got) == 0


def test_level2_handles_are_checked(ctx):
    with fe.field(fe.BABYBEAR):
        params = fe.Params(2, 0, 0, 1, 20, 2, 2)
        system = bb.System.new(ctx, params, fe.u32_add_system_inputs(), K)
        traces, claims = fe.u32_add_bench_witness(1 << 6)
        packed = fe.pack_claims(claims)
    w = system.witness(traces, packed)
    accs, s2_traces = bb.stage2_build(w, 2, (1, 2, 3, 4), (5, 6, 7, 8), (0, 0, 0, 0))
    infos = [system.circuit_info(i) for i in range(2)]
    assert [t.info() for t in s2_traces] == [(traces[i].shape[0], infos[i]["stage2_width"], 0) for i in range(2)]
    s2 = bb.pcs_commit_traces(system, params, s2_traces)
    assert s2_traces[0].info()[0] == 0                      # consumed
    with pytest.raises(pkg.MstarkError, match="consumed"):
        bb.pcs_commit_traces(system, params, s2_traces)
    with pytest.raises(pkg.MstarkError, match="LDE handle"):
        accs2, again = bb.stage2_build(w, 2, (1, 2, 3, 4), (5, 6, 7, 8), (0, 0, 0, 0))
        bb.pcs_commit_ldes(system, params, again)
    with pytest.raises(pkg.MstarkError, match="non-canonical"):
        bb.claims_accumulator(w, (P, 0, 0, 0), (1, 0, 0, 0))
    with pytest.raises(pkg.MstarkError, match="device-resident"):
        bb.commit_stage1(system.host_witness(traces, packed), params, [1 << 8])
    s1 = bb.commit_stage1(w, params, [t.shape[0] << 2 for t in traces])
    with pytest.raises(pkg.MstarkError, match="shape"):
        bb.quotient(system, 0, 6, s1, 1, s2, 1, [0] * 16, (1, 0, 0, 0))   # circuit 0 given circuit 1's matrices
    with pytest.raises(pkg.MstarkError, match="out of range"):
        bb.quotient(system, 0, 8, s1, 5, s2, 0, [0] * 16, (1, 0, 0, 0))
    # the preprocessed commitment is a view kept alive by its handle
    pre = bb.preprocessed_mmcs(system)
    assert pre is not None and np.array_equal(pre.cap, system.preprocessed_commit())
    del system
    del pre
