"""GPU parity tests, whole path: ms_prove against the oracle prover (byte-identical proofs) and the oracle
verifier (accepts; tampered proofs rejected), on the reference's own test scenarios plus the bench workload."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _prove_both(pkg, ctx, oracle, fe, inputs, params, traces, claims, lookups_from_oracle=False):
    g = pkg.System.new(ctx, params, inputs)
    o = oracle.System(g.blob)
    packed = fe.pack_claims(claims)
    assert g.preprocessed_commit() == o.preprocessed_commit()
    lookups = None
    if lookups_from_oracle:
        lookups = [o.compute_lookup_values(ci, t) for ci, t in enumerate(traces)]
    w = g.witness(traces, packed, lookups)
    proof = g.prove_multiple_claims(w).to_bytes()
    assert o.verify(packed, proof) == 0
    assert g.verify_multiple_claims(packed, proof) == 0  # the product's own verifier (ms_verify) agrees
    want = o.prove(traces, packed)
    assert proof == want
    return g, o, packed, proof


# examples/simple_proof.rs (config 1) at 4 rows and at the north-star 2^12 rows
@pytest.mark.parametrize("rows", [1, 4, 64, 4096])
def test_simple_proof(pkg, ctx, oracle, fe, rows):
    _prove_both(pkg, ctx, oracle, fe, fe.pythagorean_inputs(), fe.test_params(), [fe.pythagorean_trace(rows)], [])


# Many circuits of many heights: every distinct height is one FRI input, and those of at most 1024 values roll in inside the
# single-workgroup tail (open.hip fri_tail_k). Its table of roll-in inputs held eight until round 4's fuzzing (FUZZ_MANY, twelve
# circuits) met a system with more: heights 2^0 .. 2^11 at blow-up 2 are ten inputs below the tail's 2048 values, nine at blow-up 4
@pytest.mark.parametrize("log_blowup,final", [(1, 0), (2, 0)])
def test_many_heights_roll_into_the_fri_tail(pkg, ctx, oracle, fe, log_blowup, final):
    heights = [1 << k for k in range(12)]
    inputs = fe.pythagorean_inputs() * len(heights)
    traces = [fe.pythagorean_trace(h) for h in heights]
    params = fe.Params(log_blowup=log_blowup, cap_height=0, log_final_poly_len=final, num_queries=12, commit_proof_of_work_bits=2,
                       query_proof_of_work_bits=3)
    _prove_both(pkg, ctx, oracle, fe, inputs, params, traces, [])


def test_simple_proof_rejects_tampering(pkg, ctx, oracle, fe):
    g, o, packed, proof = _prove_both(pkg, ctx, oracle, fe, fe.pythagorean_inputs(), fe.test_params(), [fe.pythagorean_trace(8)], [])
    for pos in (len(proof) // 3, len(proof) - 40, 80):
        bad = bytearray(proof)
        bad[pos] ^= 1
        assert o.verify(packed, bytes(bad)) != 0


# src/lookup.rs:1043-1051 (even/odd lookups, claim [0,4,1], quotient degree 2)
def test_lookup_proof(pkg, ctx, oracle, fe):
    g, o, packed, proof = _prove_both(pkg, ctx, oracle, fe, fe.even_odd_inputs(), fe.test_params(), fe.even_odd_traces(), [[0, 4, 1]])
    assert o.verify(fe.pack_claims([[0, 4, 0]]), proof) != 0


# src/lookup.rs:1053-1077: a deactivated third circuit (empty trace)
def test_sparse_inactive_circuit(pkg, ctx, oracle, fe):
    traces = fe.even_odd_traces() + [np.zeros((0, 6), dtype=np.uint64)]
    _prove_both(pkg, ctx, oracle, fe, fe.even_odd_inputs(with_dead=True), fe.test_params(), traces, [[0, 4, 1]])


# src/test_circuits/u32_add.rs:193-221
def test_u32_add_proof(pkg, ctx, oracle, fe):
    traces, claims = fe.u32_add_witness([(10, 5), (30, 20), (100, 100), (8000, 10000)])
    _prove_both(pkg, ctx, oracle, fe, fe.u32_add_system_inputs(), fe.test_params(), traces, claims)


def test_u32_add_explicit_lookup_values(pkg, ctx, oracle, fe):
    traces, claims = fe.u32_add_witness([(1, 2), (0xFFFFFFFF, 1), (7, 7), (0xDEADBEEF, 0xCAFEBABE)])  # no padding rows: they would push unmatched bytes
    _prove_both(pkg, ctx, oracle, fe, fe.u32_add_system_inputs(), fe.test_params(), traces, claims, lookups_from_oracle=True)


# the bench workload (benches/multi_stark.rs) with bench_config(): 10-bit grinding, 100 queries, blowup 4
# examples/preprocessed_proof.rs: preprocessed range table + squaring circuit, one-argument lookups, no claims
@pytest.mark.parametrize("n", [16, 256])
def test_preprocessed_squares_proof(pkg, ctx, oracle, fe, n):
    _prove_both(pkg, ctx, oracle, fe, fe.squares_inputs(), fe.test_params(), fe.squares_traces(n), [])


# src/test_circuits/byte_operations.rs:124-157: wide 2^16-row preprocessed trace, lookup-only circuit, ragged claims
def test_byte_operations_proof(pkg, ctx, oracle, fe):
    traces, claims = fe.byte_operations_witness([(0, 10, 5), (1, 30, 20), (2, 100, 40), (3, 200, 100)])
    _prove_both(pkg, ctx, oracle, fe, fe.byte_operations_inputs(), fe.test_params(), traces, claims)


@pytest.mark.parametrize("log_adds", [6, 10, 14])
def test_bench_workload(pkg, ctx, oracle, fe, log_adds):
    traces, claims = fe.u32_add_bench_witness(1 << log_adds)
    _prove_both(pkg, ctx, oracle, fe, fe.u32_add_system_inputs(), fe.bench_params(), traces, claims)


# SURVEY §8d config 3 as one system on one GPU: [ByteTable, U32Add x k] — several equal-height matrices share
# every Merkle leaf and every reduced opening
@pytest.mark.parametrize("k,log_adds", [(2, 5), (8, 7)])
def test_multi_air_system(pkg, ctx, oracle, fe, k, log_adds):
    traces, claims = fe.multi_u32_add_witness(k, 1 << log_adds)
    _prove_both(pkg, ctx, oracle, fe, fe.multi_u32_add_system_inputs(k), fe.bench_params(), traces, claims)


# transforms above 2^20 take the multi-pass strided path (8-bit register pass + generic remainder)
def test_large_trace_verifies(pkg, ctx, oracle, fe):
    traces, claims = fe.u32_add_bench_witness(1 << 22)
    g = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    o = oracle.System(g.blob)
    packed = fe.pack_claims(claims)
    w = g.witness(traces, packed)
    proof = g.prove_multiple_claims(w).to_bytes()
    assert o.verify(packed, proof) == 0


def test_cap_height_and_final_poly(pkg, ctx, oracle, fe):
    params = fe.Params(log_blowup=2, cap_height=2, log_final_poly_len=2, num_queries=20, commit_proof_of_work_bits=3,
                       query_proof_of_work_bits=5)
    traces, claims = fe.u32_add_bench_witness(1 << 7)
    _prove_both(pkg, ctx, oracle, fe, fe.u32_add_system_inputs(), params, traces, claims)


def test_prove_is_deterministic_and_repeatable(pkg, ctx, oracle, fe):
    traces, claims = fe.u32_add_bench_witness(1 << 9)
    g = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    packed = fe.pack_claims(claims)
    w = g.witness(traces, packed)
    a = g.prove_multiple_claims(w).to_bytes()
    b = g.prove_multiple_claims(w, want_times=True)
    assert a == b.to_bytes()
    assert b.stage_ms["total"] > 0


# full BASELINE size (2^20 additions): size-independent properties — the oracle VERIFIER accepts the GPU proof
# (verification is cheap), the proof is reproducible, and a flipped claim is rejected.
def test_full_size_proof_verifies(pkg, ctx, oracle, fe):
    traces, claims = fe.u32_add_bench_witness(1 << 20)
    g = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    o = oracle.System(g.blob)
    packed = fe.pack_claims(claims)
    w = g.witness(traces, packed)
    proof = g.prove_multiple_claims(w).to_bytes()
    assert o.verify(packed, proof) == 0
    assert hashlib.sha256(g.prove_multiple_claims(w).to_bytes()).digest() == hashlib.sha256(proof).digest()
    bad = claims.copy()
    bad[12345, 3] ^= 1
    assert o.verify(fe.pack_claims(bad), proof) != 0


# BASELINE config 2 at its real size, byte for byte: the GPU proof of 2^20 additions equals the oracle prover's, from the
# device-resident witness and from the host-resident one (the bench's timed region: upload inside prove)
def test_full_size_proof_equals_oracle(pkg, ctx, oracle, fe):
    traces, claims = fe.u32_add_bench_witness(1 << 20)
    g = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    o = oracle.System(g.blob)
    packed = fe.pack_claims(claims)
    oracle.set_threads(16)
    try:
        want = o.prove(traces, packed)
    finally:
        oracle.set_threads(4)
    assert g.prove_multiple_claims(g.witness(traces, packed)).to_bytes() == want
    hw = g.host_witness(traces, packed)
    assert g.prove_multiple_claims(hw).to_bytes() == want
    assert g.prove_multiple_claims(hw).to_bytes() == want   # the per-proof device copies are rebuilt every time


# ms_witness_create_host: the witness stays in host memory and every proof uploads it (traces, claims) and runs
# from_stage_1 on the device; same bytes as the device-resident witness and as the oracle, on every kind of system
@pytest.mark.parametrize("case", ["bench6", "bench12", "bench12_nofusion", "byte_ops", "even_odd_dead", "squares", "pythagorean", "host_sweep"])
def test_host_resident_witness(pkg, ctx, oracle, fe, case):
    import os

    params = fe.test_params()
    if case.startswith("bench") or case == "host_sweep":
        traces, claims = fe.u32_add_bench_witness(1 << (12 if case.startswith("bench12") else 6))
        inputs, params = fe.u32_add_system_inputs(), fe.bench_params()
    elif case == "byte_ops":
        traces, claims = fe.byte_operations_witness([(0, 10, 5), (1, 30, 20), (2, 100, 40), (3, 200, 100)])
        inputs = fe.byte_operations_inputs()
    elif case == "even_odd_dead":
        traces, claims = fe.even_odd_traces() + [np.zeros((0, 6), dtype=np.uint64)], [[0, 4, 1]]
        inputs = fe.even_odd_inputs(with_dead=True)
    elif case == "squares":
        traces, claims, inputs = fe.squares_traces(64), [], fe.squares_inputs()
    else:
        traces, claims, inputs = [fe.pythagorean_trace(64)], [], fe.pythagorean_inputs()
    if case == "bench12_nofusion":   # stage 2 from materialised lookup values (from_stage_1 as its own kernel) instead of the fused kernel
        os.environ["MSAMD_NO_STAGE2_FUSION"] = "1"
    try:
        g = pkg.System.new(ctx, params, inputs)
    finally:
        os.environ.pop("MSAMD_NO_STAGE2_FUSION", None)
    packed = fe.pack_claims(claims)
    want = oracle.System(g.blob).prove(traces, packed)
    if case == "host_sweep":   # lookup values swept on the host at creation and uploaded with every proof
        os.environ["MSAMD_HOST_LOOKUP_VALUES"] = "1"
    try:
        hw = g.host_witness(traces, packed)
    finally:
        os.environ.pop("MSAMD_HOST_LOOKUP_VALUES", None)
    assert hw.pinned
    for _ in range(3):
        assert g.prove_multiple_claims(hw).to_bytes() == want
    # interleaved with a device-resident witness of the same system (pool blocks move between the two)
    dw = g.witness(traces, packed)
    assert g.prove_multiple_claims(dw).to_bytes() == want
    assert g.prove_multiple_claims(hw).to_bytes() == want
    # validation is that of ms_witness_create
    if traces[-1].size:
        bad = [t.copy() for t in traces]
        bad[-1][0, 0] = np.uint64(0xFFFFFFFF00000001)
        with pytest.raises(pkg.MstarkError, match="canonical"):
            g.host_witness(bad, packed)


# The narrow upload of a host-resident witness (traces whose values fit 1 / 2 / 4 bytes are narrowed on host threads,
# uploaded chunk by chunk and widened on the device): forced on small systems (MSAMD_PACK_MIN_BYTES=0) for every width
# class, against the plain upload (MSAMD_NO_PACK) and the oracle; a value that outgrows the width found at creation
# (the caller may rewrite its buffers between proofs) must send the proof down the plain path, not corrupt it
@pytest.mark.parametrize("case", ["bytes", "u16", "u32", "wide", "bench12"])
def test_host_resident_witness_narrow_upload(pkg, ctx, oracle, fe, case, monkeypatch):
    monkeypatch.setenv("MSAMD_PACK_MIN_BYTES", "0")
    if case == "bench12":
        traces, claims = fe.u32_add_bench_witness(1 << 12)
        inputs, params = fe.u32_add_system_inputs(), fe.bench_params()
    else:
        # the Pythagorean circuit a^2 + b^2 = c^2 takes any field elements: rows scaled to the width class under test
        top = {"bytes": 1 << 8, "u16": 1 << 16, "u32": 1 << 32, "wide": 1 << 63}[case]
        t = fe.pythagorean_trace(64).copy()
        assert int(t.max()) < 256
        if case != "bytes":
            # (3k, 4k, 5k) stays a Pythagorean triple: scale one row so that the largest value lands in the class
            k = (top - 1) // int(t[1].max())
            t[1] = t[1] * np.uint64(k)
        traces, claims, inputs, params = [np.ascontiguousarray(t, dtype=np.uint64)], [], fe.pythagorean_inputs(), fe.test_params()
    g = pkg.System.new(ctx, params, inputs)
    packed = fe.pack_claims(claims)
    want = oracle.System(g.blob).prove(traces, packed)
    hw = g.host_witness(traces, packed)
    for _ in range(3):
        assert g.prove_multiple_claims(hw).to_bytes() == want
    monkeypatch.setenv("MSAMD_NO_PACK", "1")
    plain = g.host_witness(traces, packed)
    assert g.prove_multiple_claims(plain).to_bytes() == want
    monkeypatch.delenv("MSAMD_NO_PACK")
    if case in ("bytes", "u16", "u32"):
        # the caller rewrites its buffer between proofs: a wider value, then back
        kept = hw.keep[0]
        old = kept[5].copy()
        kept[5] = old * np.uint64(1 << 40)   # still a Pythagorean triple, far outside the width class
        t2 = [kept.copy()]
        got = g.prove_multiple_claims(hw).to_bytes()
        assert got == oracle.System(g.blob).prove(t2, packed)
        kept[5] = old
        assert g.prove_multiple_claims(hw).to_bytes() == want


# Row groups (MSAMD_ROW_GROUPS=1; measured slower on the pool's hosts, hence opt-in): from 2^19 rows on the narrow upload of the plain
# prover is cut into eight groups of rows that each complete whole tiles of the inverse transform's first pass
# (HostUpload::row_groups), and that pass runs group by group as the rows arrive. Odd width, 16- and 32-row runs, every width
# class; a value that outgrows its class in the FIRST group and one in the SIXTH (five groups already transposed and
# transformed: abandoned) must give the plain path's proof
@pytest.mark.parametrize("log_h,case", [(19, "bytes"), (19, "u16"), (20, "u32")])
def test_host_resident_witness_row_groups(pkg, ctx, oracle, fe, log_h, case, monkeypatch):
    monkeypatch.setenv("MSAMD_ROW_GROUPS", "1")
    top = {"bytes": 1 << 8, "u16": 1 << 16, "u32": 1 << 32}[case]
    h = 1 << log_h
    t = fe.pythagorean_trace(h).copy()
    # (k a, k b, k c) stays a triple: a row-dependent factor makes the rows differ (a row landing in another row's place must show)
    kmax = (top - 1) // int(t.max())
    k = (np.arange(h, dtype=np.uint64) * np.uint64(2654435761) >> np.uint64(7)) % np.uint64(kmax) + np.uint64(1)
    t = t * k[:, None]
    t[1] = fe.pythagorean_trace(4)[1] * np.uint64(kmax)   # the class's largest value is present
    assert top // 2 <= int(t.max()) < top or case == "bytes"
    traces = [np.ascontiguousarray(t, dtype=np.uint64)]
    g = pkg.System.new(ctx, fe.test_params(), fe.pythagorean_inputs())
    packed = fe.pack_claims([])
    osys = oracle.System(g.blob)
    want = osys.prove(traces, packed)
    hw = g.host_witness(traces, packed)
    for _ in range(2):
        assert g.prove_multiple_claims(hw).to_bytes() == want
    # the path under test is the one that ran: eight transposes and eight first passes for the one matrix
    ctx.set_profile(["transpose", "ntt12_dit"])
    ctx.reset_stats()
    assert g.prove_multiple_claims(hw).to_bytes() == want
    st = ctx.kernel_stats()
    ctx.set_profile([])
    assert st["transpose"]["launches"] == 8 and st["ntt12_dit"]["launches"] >= 8, st
    assert g.prove_multiple_claims(g.witness(traces, packed)).to_bytes() == want
    kept = hw.keep[0]
    run = 1 << (log_h - 12 - 3)
    for row in (5, 5 * run + 3, h - 1):   # groups 0, 5 and 7
        old = kept[row].copy()
        kept[row] = old * np.uint64(1 << 20)   # still a triple, outside every width class's range or at least this one's
        assert int(kept[row].max()) >= top
        assert g.prove_multiple_claims(hw).to_bytes() == osys.prove([kept.copy()], packed), row
        kept[row] = old
        assert g.prove_multiple_claims(hw).to_bytes() == want


# A worker of the narrowing pool that is descheduled inside its piece (shared host cores) must not hold its chunk up: once every
# piece of a chunk has a taker, pieces still in work after 30 us are narrowed again by the calling thread, and late workers
# leave on their own (the next job and the witness's destructor wait for them). MSAMD_PACK_STRAGGLE_US holds one piece per
# chunk back on its worker: same bytes, and a proof does not pay the eight sleeps
def test_late_pack_workers_do_not_hold_the_upload(pkg, ctx, oracle, fe, monkeypatch):
    import time

    traces, claims = fe.u32_add_bench_witness(1 << 16)   # (the adders' 7 MB trace is narrowed, the byte table's is below the threshold: one job per proof)
    g = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    packed = fe.pack_claims(claims)
    hw = g.host_witness(traces, packed)
    want = g.prove_multiple_claims(g.witness(traces, packed)).to_bytes()
    assert g.prove_multiple_claims(hw).to_bytes() == want
    t0 = time.perf_counter()
    for _ in range(3):
        assert g.prove_multiple_claims(hw).to_bytes() == want
    plain_ms = (time.perf_counter() - t0) / 3 * 1e3
    monkeypatch.setenv("MSAMD_PACK_STRAGGLE_US", "20000")
    t0 = time.perf_counter()
    assert g.prove_multiple_claims(hw).to_bytes() == want   # (the FIRST proof under it: later ones wait for this one's sleepers to leave)
    late_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(2):
        assert g.prove_multiple_claims(hw).to_bytes() == want
    monkeypatch.delenv("MSAMD_PACK_STRAGGLE_US")
    assert g.prove_multiple_claims(hw).to_bytes() == want
    assert late_ms < plain_ms + 10.0, (plain_ms, late_ms)   # one sleep is 20 ms, eight chunks would have waited for eight
    del hw   # (the destructor waits for any worker still inside a piece)


# Two host-resident witnesses made from the SAME buffers share one page lock, counted per range: when the first one goes, the
# second must still upload from locked memory (a second hipHostRegister of a range only reports "already registered"; before
# round 4's count the first witness's unregister left the second one's asynchronous copies reading memory the GPU could no longer
# see - the kind of access that ended a fuzzing run with a GPU memory fault). Both configurations.
def test_two_host_witnesses_share_their_buffers(pkg, ctx, oracle, fe):
    import gc

    traces, claims = fe.u32_add_bench_witness(1 << 12)
    g = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    packed = fe.pack_claims(claims)
    want = oracle.System(g.blob).prove(traces, packed)
    first = g.host_witness(traces, packed)
    second = g.host_witness(traces, packed)
    assert all(a is b for a, b in zip(first.keep, second.keep))   # (the same arrays, not copies)
    assert g.prove_multiple_claims(first).to_bytes() == want
    del first
    gc.collect()
    for _ in range(3):
        assert g.prove_multiple_claims(second).to_bytes() == want
    third = g.host_witness(traces, packed)   # and a new owner while the second still holds the lock
    del second
    gc.collect()
    assert g.prove_multiple_claims(third).to_bytes() == want
    bb = pkg.babybear
    import oracle_bb as ob

    K = fe.poseidon2_constants()
    with fe.field(fe.BABYBEAR):
        bs = bb.System.new(ctx, fe.test_params(), fe.mul_air_inputs(), K)
        tr = [fe.mul_air_trace(1 << 10)]
        none = fe.pack_claims([])
    ob.set_poseidon2(K)
    bwant = ob.System(bs.blob).prove(tr, none)
    tr32 = [np.ascontiguousarray(tr[0], dtype=np.uint32)]   # (the witness keeps an array of this type as it is: both share it)
    b1 = bs.host_witness(tr32, none)
    b2 = bs.host_witness(tr32, none)
    assert b1.keep[0] is tr32[0] and b2.keep[0] is tr32[0]
    assert bs.prove_multiple_claims(b1).to_bytes() == bwant
    del b1
    gc.collect()
    assert bs.prove_multiple_claims(b2).to_bytes() == bwant and bs.prove_multiple_claims(b2).to_bytes() == bwant


# ms_witness_prefetch: each proof of a host-resident witness also uploads the inputs of the next one; same bytes, also
# across switching it on and off, interleaved with other witnesses, and after an injected mid-proof failure
def test_host_witness_prefetch(pkg, ctx, oracle, fe):
    traces, claims = fe.u32_add_bench_witness(1 << 12)
    g = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    packed = fe.pack_claims(claims)
    want = oracle.System(g.blob).prove(traces, packed)
    hw = g.host_witness(traces, packed)
    dw = g.witness(traces, packed)
    with pytest.raises(pkg.MstarkError):
        dw.prefetch(True)
    assert g.prove_multiple_claims(hw).to_bytes() == want
    hw.prefetch(True)
    for _ in range(4):
        assert g.prove_multiple_claims(hw).to_bytes() == want
    assert g.prove_multiple_claims(dw).to_bytes() == want
    assert g.prove_multiple_claims(hw).to_bytes() == want
    ctx.debug_fail_alloc(25)
    with pytest.raises(pkg.MstarkError, match="injected"):
        g.prove_multiple_claims(hw)
    ctx.debug_fail_alloc(0)
    assert g.prove_multiple_claims(hw).to_bytes() == want
    hw.prefetch(False)
    assert g.prove_multiple_claims(hw).to_bytes() == want
    hw.prefetch(True)
    assert g.prove_multiple_claims(hw).to_bytes() == want
    del hw   # a prefetch is pending: the destructor waits for it


# BASELINE config 5: 2^26 additions (228 GiB of the 288 GiB HBM; about a minute with witness generation and the oracle
# verifier). Runs by default; MSAMD_SKIP_STRESS=1 leaves it out on a box that is short of memory. The recorded runs are
# profiles/r01_config5_stress.txt and profiles/r02_config5.txt (tools/stress.py is the same flow as a script).
@pytest.mark.skipif(bool(__import__("os").environ.get("MSAMD_SKIP_STRESS")), reason="MSAMD_SKIP_STRESS set")
def test_config5_two_pow_26_verifies(pkg, ctx, oracle, fe):
    ctx.trim()
    traces, claims = fe.u32_add_bench_witness(1 << 26)
    g = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    o = oracle.System(g.blob)
    packed = fe.pack_claims(claims)
    w = g.witness(traces, packed)
    proof = g.prove_multiple_claims(w).to_bytes()
    assert o.verify(packed, proof) == 0
    assert g.verify(packed, proof) == 0
    bad = claims.copy()
    bad[54321, 2] ^= 1
    assert o.verify(fe.pack_claims(bad), proof) != 0
    del w
    ctx.trim()


# config 5's shape at 2^24 additions, byte for byte against the oracle prover (2^26-row LDEs: the multi-pass transforms,
# 26-level trees, 24 FRI rounds of which the first ones exceed the fused kernels' sizes)
@pytest.mark.skipif(bool(__import__("os").environ.get("MSAMD_SKIP_STRESS")), reason="MSAMD_SKIP_STRESS set")
def test_two_pow_24_proof_equals_oracle(pkg, ctx, oracle, fe):
    traces, claims = fe.u32_add_bench_witness(1 << 24)
    g = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    packed = fe.pack_claims(claims)
    w = g.witness(traces, packed)
    proof = g.prove_multiple_claims(w).to_bytes()
    del w
    ctx.trim()
    oracle.set_threads(16)   # the suite's default pool of 4 would take six minutes here
    try:
        want = oracle.System(g.blob).prove(traces, packed)
    finally:
        oracle.set_threads(4)
    assert proof == want


# The shape of the reference's largest scenario (src/test_circuits/blake3.rs:2215-2613: a ten-circuit system whose main
# circuit has a 2625-column trace, next to tiny table circuits): a synthetic AIR with 2700 columns and about 8000 nodes -
# above the hiprtc path's 3000-node limit, so the interpreter with its slot file in global memory runs - linked by
# lookups to two small circuits of other heights.
def test_wide_air_large_program(pkg, ctx, oracle, fe):
    P = fe.P
    W, H = 2700, 64
    rng = np.random.default_rng(77)
    wide = np.zeros((H, W), dtype=object)
    base = int(rng.integers(1, 1 << 20))
    wide[:, 0] = [base + r for r in range(H)]   # next row = this row + 1
    wide[:, 1] = [int(x) for x in rng.integers(1, 1 << 20, H)]
    for i in range(2, W):
        wide[:, i] = (wide[:, i - 2] * wide[:, i - 1]) % P
    wide = wide.astype(np.uint64)

    def ev(b):
        local, nxt = b.main()
        for i in range(2, W):
            b.assert_eq(local[i - 2] * local[i - 1], local[i])
        b.when_transition().assert_eq(nxt[0], local[0] + fe.Expr.const(1))

    var = fe.Expr.main
    one = fe.Expr.const(1)
    wide_air = fe.lookup_air(W, ev, [fe.Lookup.push(one, [var(0), var(1)]), fe.Lookup.push(one, [var(W - 2), var(W - 1)])])
    # two small tables of other heights pull what the wide circuit pushes
    t1 = np.stack([wide[:, 0], wide[:, 1]], axis=1)
    t2 = np.zeros((128, 3), dtype=np.uint64)
    t2[:H, 0] = 1
    t2[:H, 1] = wide[:, W - 2]
    t2[:H, 2] = wide[:, W - 1]
    a1 = fe.lookup_air(2, None, [fe.Lookup.pull(one, [var(0), var(1)])])
    a2 = fe.lookup_air(3, lambda b: b.assert_bool(b.main()[0][0]), [fe.Lookup.pull(var(0), [var(1), var(2)])])
    g, o, packed, proof = _prove_both(pkg, ctx, oracle, fe, [a1, wide_air, a2], fe.test_params(), [t1, wide, t2], [])
    info = g.circuit_info(1)
    assert info["main_width"] == W and info["constraint_count"] > W


# ms_verify against the oracle verifier: same verdict on the untouched proof, on wrong claims and on corrupted bytes
# all over the proof (commitments, accumulators, FRI data, opened values)
def test_product_verifier_agrees_with_oracle(pkg, ctx, oracle, fe):
    traces, claims = fe.u32_add_bench_witness(1 << 9)
    g, o, packed, proof = _prove_both(pkg, ctx, oracle, fe, fe.u32_add_system_inputs(), fe.bench_params(), traces, claims)
    bad = claims.copy()
    bad[7, 2] ^= 1
    assert g.verify(fe.pack_claims(bad), proof) == o.verify(fe.pack_claims(bad), proof) != 0
    assert g.verify(fe.pack_claims(claims[:-1]), proof) != 0
    rng = np.random.default_rng(5)
    n = len(proof)
    spots = [8, 10 + 8 + 5, 10 + 48 + 8 + 3, 150, n - 5, n // 2, n // 3] + [int(x) for x in rng.integers(0, n, 40)]
    rejected = 0
    for pos in spots:
        t = bytearray(proof)
        t[pos] ^= 1 << int(rng.integers(0, 8))
        a, b = g.verify(packed, bytes(t)), o.verify(packed, bytes(t))
        assert (a == 0) == (b == 0), (pos, a, b)
        rejected += a != 0
    assert rejected == len(spots)  # every single-bit corruption of these proofs is caught
    assert g.verify(packed, proof[:-1]) == 3 and g.verify(packed, proof + b"\x00") == 3  # InvalidProofShape
    assert g.verify(packed, b"") == 3


# verification at the full bench size: the claims (42 MB of transcript, 2^20 inversions) are handled on the device
def test_product_verifier_full_size(pkg, ctx, oracle, fe):
    traces, claims = fe.u32_add_bench_witness(1 << 20)
    g = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    packed = fe.pack_claims(claims)
    proof = g.prove_multiple_claims(g.witness(traces, packed)).to_bytes()
    assert g.verify(packed, proof) == 0
    bad = claims.copy()
    bad[99999, 1] ^= 4
    assert g.verify(fe.pack_claims(bad), proof) != 0


# the bench witness generated in HBM (ms_witness_u32_add_bench) must give the proof of the host-built witness, also for
# a size that needs padding rows and for other seeds
@pytest.mark.parametrize("num_adds,a0,b0", [(1 << 10, 0xDEADBEEF, 0xCAFEBABE), (1000, 0xDEADBEEF, 0xCAFEBABE), (1 << 14, 0x12345678, 0x9ABCDEF1),
                                            (1, 7, 9)])
def test_device_generated_bench_witness(pkg, ctx, oracle, fe, num_adds, a0, b0):
    g = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    traces, claims = fe.u32_add_bench_witness(num_adds, a0, b0)
    packed = fe.pack_claims(claims)
    want = g.prove_multiple_claims(g.witness(traces, packed)).to_bytes()
    wd = g.bench_witness_on_device(num_adds, a0, b0)
    assert wd.rows == 256 + traces[1].shape[0]
    got = g.prove_multiple_claims(wd).to_bytes()
    assert got == want
    assert got == oracle.System(g.blob).prove(traces, packed)   # the checker is the oracle prover, not only the product
    # padding rows push twelve zero bytes nobody pulls (true of the reference's generator too): only full traces balance
    assert g.verify(packed, got) == (0 if num_adds & (num_adds - 1) == 0 else 6)


# every alternative code path selectable by environment variable must give the same proof bytes: host-driven FRI rounds,
# host-side query step, no single-workgroup FRI tail, host sweep for the lookup values, interpreter kernels instead of
# the hiprtc-compiled ones (the library reads these variables at call time)
def test_device_side_opening_step(pkg, ctx, oracle, fe, monkeypatch, capfd):
    """MSAMD_DEV_OPENING=1: the opened values' transcript step (finishing factors, absorption, the FRI batching challenge, the
    reduced openings' coefficients and constants) as one launch, FRI continuing from the device state: ONE host wait per proof,
    the same bytes. Systems: the bench workload (two points per matrix, a short circuit on the side stream), the nine-circuit
    system, a preprocessed table with lookups and claims, host-resident witnesses."""
    import numpy as np

    cases = []
    tr, cl = fe.u32_add_bench_witness(1 << 14)
    cases.append((fe.u32_add_system_inputs(), fe.bench_params(), tr, cl))
    tr, cl = fe.multi_u32_add_witness(8, 1 << 11)
    cases.append((fe.multi_u32_add_system_inputs(8), fe.bench_params(), tr, cl))
    rng = np.random.default_rng(5)
    calls = [(int(rng.integers(0, 4)), int(rng.integers(0, 256)), int(rng.integers(0, 256))) for _ in range(3000)]
    tr, cl = fe.byte_operations_witness(calls)
    cases.append((fe.byte_operations_inputs(), fe.test_params(), tr, cl))
    for inputs, params, tr, cl in cases:
        packed = fe.pack_claims(cl)
        g = pkg.System.new(ctx, params, inputs)
        want = oracle.System(g.blob).prove(tr, packed)
        w, hw = g.witness(tr, packed), g.host_witness(tr, packed)
        assert g.prove_multiple_claims(w).to_bytes() == want
        monkeypatch.setenv("MSAMD_DEV_OPENING", "1")
        n0 = ctx.sync_count()
        assert g.prove_multiple_claims(w).to_bytes() == want
        assert ctx.sync_count() - n0 == 1, "the device-side opening step leaves ONE host wait per proof"
        assert g.prove_multiple_claims(hw).to_bytes() == want and g.prove_multiple_claims(w).to_bytes() == want
        monkeypatch.delenv("MSAMD_DEV_OPENING")
        n0 = ctx.sync_count()
        assert g.prove_multiple_claims(w).to_bytes() == want
        assert ctx.sync_count() - n0 == 2


@pytest.mark.parametrize("var", ["MSAMD_HOST_FRI", "MSAMD_HOST_QUERY", "MSAMD_NO_FRI_TAIL", "MSAMD_HOST_LOOKUP_VALUES", "MSAMD_NO_JIT", "MSAMD_NO_SUBTREE",
                                 "MSAMD_MATERIALISE_LOOKUPS", "MSAMD_NO_FRI_FUSED", "MSAMD_NO_FLAG_SYNC", "MSAMD_NO_SIDE_STREAM", "MSAMD_OLD_TRANSPOSE", "MSAMD_GENERIC_LEAF_HASH", "MSAMD_NO_DEEP_LEAVES",
                                 "MSAMD_HOST_TRANSCRIPT", "MSAMD_NO_NEXT_SHIFT", "MSAMD_OLD_CLAIMS_TREE", "MSAMD_CLAIMS_ACC_MAIN", "MSAMD_NO_BARY_BATCH", "MSAMD_NO_NTT_SMALL",
                                 "MSAMD_NO_WIDE_PREHASH", "MSAMD_NO_WAVE_QUOTIENT"])
def test_alternative_paths_give_the_same_proof(pkg, ctx, oracle, fe, var):
    import os

    traces, claims = fe.u32_add_bench_witness(1 << 13)   # 2^15-row LDEs: FRI rounds above and inside the tail kernel
    packed = fe.pack_claims(claims)
    inputs, params = fe.u32_add_system_inputs(), fe.bench_params()
    g = pkg.System.new(ctx, params, inputs)
    want = g.prove_multiple_claims(g.witness(traces, packed)).to_bytes()
    assert oracle.System(g.blob).verify(packed, want) == 0
    assert var not in os.environ
    os.environ[var] = "1"
    try:
        g2 = pkg.System.new(ctx, params, inputs)           # MSAMD_NO_JIT acts at System::new
        got = g2.prove_multiple_claims(g2.witness(traces, packed)).to_bytes()
    finally:
        del os.environ[var]
    assert got == want


# The outer transcript on the device (csrc/outer.hip): with more than 8192 claim words the claims digest is computed on the
# device and beta/gamma, alpha, zeta are sampled there; the host challenger replays and checks. Shapes that exercise every
# piece: several circuits of one height (one next-row point for all of them), a preprocessed commitment (a fourth opening
# round on symbolic points), a Merkle cap of several digests inside the hashed pieces, few long claims (the claims sum on the
# device with fewer than 256 claims). Each is compared with the oracle and with the host-transcript path, and the probe log
# must show that the device path actually ran.
@pytest.mark.parametrize("case", ["multi_air", "preprocessed", "cap4", "long_claims"])
def test_device_transcript(pkg, ctx, oracle, fe, case, monkeypatch, capfd):
    import numpy as np

    if case == "multi_air":
        inputs, params = fe.multi_u32_add_system_inputs(8), fe.bench_params()
        traces, claims = fe.multi_u32_add_witness(8, 1 << 11)
    elif case == "preprocessed":
        rng = np.random.default_rng(5)
        calls = [(int(rng.integers(0, 4)), int(rng.integers(0, 256)), int(rng.integers(0, 256))) for _ in range(3000)]
        inputs, params = fe.byte_operations_inputs(), fe.test_params()
        traces, claims = fe.byte_operations_witness(calls)
    elif case == "cap4":
        inputs = fe.u32_add_system_inputs()
        params = fe.Params(log_blowup=2, cap_height=2, log_final_poly_len=2, num_queries=20, commit_proof_of_work_bits=3, query_proof_of_work_bits=5)
        traces, claims = fe.u32_add_bench_witness(1 << 12)
    else:
        # four claims of 3000 words on top of a balanced system: no circuit pushes them, so the verifier rejects the proof,
        # but the prover does not check the balance - the bytes are compared with the oracle's without a verdict
        inputs, params = fe.u32_add_system_inputs(), fe.test_params()
        traces, claims = fe.u32_add_bench_witness(1 << 6)
        rng = np.random.default_rng(9)
        claims = list(claims) + [[int(v) for v in rng.integers(0, 2**62, size=3000)] for _ in range(4)]
    packed = fe.pack_claims(claims)
    g = pkg.System.new(ctx, params, inputs)
    o = oracle.System(g.blob)
    want = o.prove(traces, packed) if case == "long_claims" else None
    monkeypatch.setenv("MSAMD_TRACE_HOST", "1")
    capfd.readouterr()
    got = g.prove_multiple_claims(g.witness(traces, packed)).to_bytes()
    err = capfd.readouterr().err
    assert "transcript replayed" in err and "sync 2 (stage-2 cap)" not in err, err[-2000:]
    monkeypatch.delenv("MSAMD_TRACE_HOST")
    if want is None:
        want = o.prove(traces, packed)
        assert o.verify(packed, got) == 0
    assert got == want
    monkeypatch.setenv("MSAMD_HOST_TRANSCRIPT", "1")
    assert g.prove_multiple_claims(g.witness(traces, packed)).to_bytes() == want


# handles may die in any order (a garbage collector gives none): the library keeps a context alive while systems or
# witnesses built on it exist
def test_handles_survive_out_of_order_destruction(pkg, fe):
    import gc

    c = pkg.Context(0)
    g = pkg.System.new(c, fe.bench_params(), fe.u32_add_system_inputs())
    w = g.bench_witness_on_device(1 << 8)
    want = g.prove_multiple_claims(w).to_bytes()
    c.close()                      # the owner lets go of the context first
    assert g.prove_multiple_claims(w).to_bytes() == want
    h = w.h
    w.h = None                     # detach so that the wrapper's own ordering cannot help
    sysh = g.h
    g.h = None
    pkg.lib().ms_system_destroy(sysh)   # system before its witness
    pkg.lib().ms_witness_destroy(h)     # last dependent: system and context are released here
    gc.collect()


# the reference's panics become error codes with a message (include/mstark.h), and the library stays usable afterwards:
# src/prover.rs:323-326 (all circuits inactive), src/system.rs:249-264 (height mismatches), src/system.rs:171-178
# (constraint degree above the blow-up), malformed blobs, non-canonical inputs, a too-small proof buffer
def test_error_paths_return_codes_not_crashes(pkg, ctx, fe):
    inputs, params = fe.u32_add_system_inputs(), fe.bench_params()
    g = pkg.System.new(ctx, params, inputs)
    traces, claims = fe.u32_add_bench_witness(1 << 6)
    packed = fe.pack_claims(claims)
    want = g.prove_multiple_claims(g.witness(traces, packed)).to_bytes()
    # every circuit inactive
    empty = [np.zeros((0, 1), dtype=np.uint64), np.zeros((0, 14), dtype=np.uint64)]
    with pytest.raises(pkg.MstarkError, match="deactivated"):
        g.prove_multiple_claims(g.witness(empty, fe.pack_claims([])))
    # preprocessed circuit with a trace of the wrong height; height not a power of two
    with pytest.raises(pkg.MstarkError, match="preprocessed"):
        g.witness([np.zeros((128, 1), dtype=np.uint64), traces[1]], packed)
    with pytest.raises(pkg.MstarkError, match="power of two"):
        g.witness([traces[0], traces[1][:48]], packed)
    # non-canonical field elements in traces and claims
    bad = traces[1].copy()
    bad[3, 2] = np.uint64(0xFFFFFFFF00000001)
    with pytest.raises(pkg.MstarkError, match="canonical"):
        g.witness([traces[0], bad], packed)
    offs, data = packed
    badc = data.copy()
    badc[5] = np.uint64(0xFFFFFFFFFFFFFFFF)
    with pytest.raises(pkg.MstarkError, match="canonical"):
        g.witness(traces, (offs, badc))
    # malformed system blobs
    blob = g.blob
    for mutilated in (blob[:40], b"\x00" * 64, blob[:-8]):
        with pytest.raises(pkg.MstarkError):
            pkg.System(ctx, mutilated, 2)
    # x^5 = y needs quotient degree 4 > blow-up 2 (src/system.rs:404-445)
    def ev(b):
        local, _ = b.main()
        x = local[0]
        b.assert_eq(x * x * x * x * x, local[1])
    with pytest.raises(pkg.MstarkError, match="degree"):
        pkg.System.new(ctx, fe.test_params(), [fe.lookup_air(2, ev, [])])
    # a proof buffer that is too small is reported with the needed size and the call is repeated
    g._proof_cap = 1000
    assert g.prove_multiple_claims(g.witness(traces, packed)).to_bytes() == want
    assert g._proof_cap == len(want)
    # and nothing above left the library in a bad state
    assert g.prove_multiple_claims(g.witness(traces, packed)).to_bytes() == want


# An error in the middle of a proof (here: an injected allocation failure at every allocation in turn) must return an
# error code, drop the read-backs it had queued into locals that no longer exist, and leave the library usable.
# (side = "9": the byte table's launches go to the side stream beside the adder's - failures then also unwind out of side
# scopes, with blocks of the side pool live and releases deferred)
@pytest.mark.parametrize("side", ["", "9"])
@pytest.mark.parametrize("host", [False, True])
def test_mid_proof_failure_leaves_the_library_usable(pkg, ctx, fe, host, side, monkeypatch):
    if side:
        monkeypatch.setenv("MSAMD_SIDE_MAX_LOG", side)
    traces, claims = fe.u32_add_bench_witness(1 << 11)   # claims hashed on the device: read-backs are queued mid-proof
    g = pkg.System.new(ctx, fe.bench_params(), fe.u32_add_system_inputs())
    packed = fe.pack_claims(claims)
    w = g.host_witness(traces, packed) if host else g.witness(traces, packed)
    want = g.prove_multiple_claims(w).to_bytes()
    failures = 0
    for nth in range(1, 400):
        ctx.debug_fail_alloc(nth)
        try:
            got = g.prove_multiple_claims(w).to_bytes()
        except pkg.MstarkError as e:
            assert "injected" in str(e)
            failures += 1
            ctx.debug_fail_alloc(0)
            assert g.prove_multiple_claims(w).to_bytes() == want
            continue
        ctx.debug_fail_alloc(0)
        assert got == want     # the proof needed fewer than nth allocations
        break
    assert failures > 20


# a seeded slice of tools/fuzz_parity.py: random systems (constraint graphs, lookups, preprocessed traces, inactive
# circuits, ragged claims, PCS / FRI parameters); identical proof bytes and agreeing verifiers
def test_random_systems_differential(pkg, ctx, oracle, fe):
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity

    os.environ["MSAMD_NO_JIT"] = "1"   # a new circuit per case: skip the hiprtc compile, the interpreter is the subject
    try:
        rng = np.random.default_rng(2026)
        tally = {}
        for case in range(120):
            r = fuzz_parity.one_case(pkg, fe, oracle, ctx, np.random.default_rng(rng.integers(0, 1 << 62)), case)
            tally[r] = tally.get(r, 0) + 1
    finally:
        del os.environ["MSAMD_NO_JIT"]
    assert tally.get("proved", 0) + tally.get("verified", 0) >= 60, tally


# the side stream (short circuits' launches queued beside the long circuits' - prover.hip) with the threshold pulled down
# so that the small random systems split into "short" and "long" circuits in every possible way
@pytest.mark.parametrize("max_log", ["2", "4", "6"])
def test_random_systems_differential_side_stream(pkg, ctx, oracle, fe, max_log):
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity

    os.environ["MSAMD_NO_JIT"] = "1"
    os.environ["MSAMD_SIDE_MAX_LOG"] = max_log
    try:
        rng = np.random.default_rng(4052 + int(max_log))
        tally = {}
        for case in range(60):
            r = fuzz_parity.one_case(pkg, fe, oracle, ctx, np.random.default_rng(rng.integers(0, 1 << 62)), case)
            tally[r] = tally.get(r, 0) + 1
    finally:
        del os.environ["MSAMD_NO_JIT"]
        del os.environ["MSAMD_SIDE_MAX_LOG"]
    assert tally.get("proved", 0) + tally.get("verified", 0) >= 30, tally
