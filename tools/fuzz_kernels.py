"""Differential fuzzing of the PCS-level entry points (the pieces a host keeping its own prover loop would call):
random shapes for ms_dft_batch, ms_coset_lde_batch, ms_quotient_lde, ms_mmcs_commit/open, ms_stage2_trace,
ms_claims_accumulator and ms_blake3, each compared bit for bit with the oracle.

usage: python3 tools/fuzz_kernels.py [N_CASES_PER_ENTRY_POINT] [SEED]
       FUZZ_FIELD=babybear python3 tools/fuzz_kernels.py ...   the msbb_* entry points (include/mstark_bb.h) against
       oracle/libms_oracle_bb.so: transforms, coset LDE, Poseidon2 permutation, Merkle commit / open
The oracle is used only as the checker."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402

P = 0xFFFFFFFF00000001


def rand_field(rng, shape):
    v = rng.integers(0, P, shape, dtype=np.uint64)
    edge = np.array([0, 1, 2, P - 1, P - 2, (1 << 32) - 1, 1 << 32, (1 << 32) + 1], dtype=np.uint64)
    mask = rng.random(shape) < 0.1
    return np.where(mask, edge[rng.integers(0, len(edge), shape)], v)


def main_babybear(n, seed):
    pkg = load_package()
    import oracle_bb as ob

    bb, fe = pkg.babybear, pkg.frontend
    PB = fe.BABYBEAR["P"]
    ctx = pkg.Context(0)
    K = fe.poseidon2_constants(seed)
    bb.set_poseidon2(ctx, K)
    ob.set_poseidon2(K)
    rng = np.random.default_rng(seed)
    t0 = time.time()

    def rf(shape):
        v = rng.integers(0, PB, shape, dtype=np.uint64)
        edge = np.array([0, 1, 2, PB - 1, PB - 2, 1 << 27, (1 << 27) + 1, 1 << 30], dtype=np.uint64)
        return np.where(rng.random(shape) < 0.1, edge[rng.integers(0, len(edge), shape)], v)

    def done(what, k):
        print("[fuzz %6.1fs] %-28s %d cases identical" % (time.time() - t0, what, k), flush=True)

    for _ in range(n):
        log_h = int(rng.choice([0, 1, 2, 3, 4, 6, 7, 9, 11, 12, 13, 14, 15, 17, 18, 19, 20, 21]))
        m = rf((1 << log_h, int(rng.integers(1, 4 if log_h > 16 else 30))))
        inv = bool(rng.integers(0, 2))
        assert np.array_equal(bb.dft_batch(ctx, m, inverse=inv), ob.dft_batch(m, inverse=inv)), ("dft", log_h, inv)
    done("msbb_dft_batch", n)
    for _ in range(n):
        log_h, lb = int(rng.choice([0, 1, 2, 3, 5, 8, 10, 12, 13, 14, 16, 17])), int(rng.integers(1, 4))
        m = rf((1 << log_h, int(rng.integers(1, 4 if log_h > 14 else 30))))
        assert np.array_equal(bb.coset_lde_batch(ctx, m, lb), ob.coset_lde_bitrev(m, lb)), ("lde", log_h, lb)
    done("msbb_coset_lde_batch", n)
    for _ in range(n):
        st = rf((int(rng.integers(1, 40)), 16))
        got = bb.poseidon2_permute(ctx, st)
        assert all(np.array_equal(got[i], ob.poseidon2_permute(st[i])) for i in range(st.shape[0]))
    done("msbb_poseidon2_permute", n)
    for _ in range(n):  # 1-6 matrices, mixed heights up to 2^17 (one-lane, 16-lane and one-launch layers; injections), wide rows, caps
        nm = int(rng.integers(1, 7))
        top = int(rng.choice([3, 8, 12, 16, 17]))
        shapes = [(1 << int(rng.integers(0, top + 1)), int(rng.integers(1, 120 if rng.random() < 0.15 else 20))) for _ in range(nm)]
        shapes = [(h, w if h <= 4096 else min(w, 12)) for h, w in shapes]
        cap_h = int(rng.integers(0, 4))
        mats = [rf(sh) for sh in shapes]
        g, o = bb.Mmcs(ctx, mats, cap_h), ob.Mmcs(mats, cap_h)
        assert np.array_equal(g.cap, np.frombuffer(o.cap, dtype=np.uint32)), ("mmcs cap", shapes, cap_h)
        maxh = max(sh[0] for sh in shapes)
        for index in {0, maxh - 1, int(rng.integers(0, maxh))}:
            gv, gp = g.open(index)
            ov, op = o.open(index)
            assert np.array_equal(gv, ov) and np.array_equal(gp, np.frombuffer(op, dtype=np.uint32)), ("mmcs open", shapes, cap_h, index)
    done("msbb_mmcs_commit / open", n)
    print("OK: %d random cases per BabyBear entry point, every result identical to the oracle's" % n)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if os.environ.get("FUZZ_FIELD", "") == "babybear":
        return main_babybear(n, seed)
    pkg = load_package()
    import oracle

    ctx = pkg.Context(0)
    rng = np.random.default_rng(seed)
    t0 = time.time()

    def done(what, k):
        print("[fuzz %6.1fs] %-28s %d cases identical" % (time.time() - t0, what, k), flush=True)

    for _ in range(n):  # transforms: every size class of the pass planner (generic, 12-bit, 8+12, multi-pass)
        log_h = int(rng.choice([0, 1, 2, 3, 4, 6, 7, 9, 11, 12, 13, 15, 17, 18, 19, 20, 21]))
        w = int(rng.integers(1, 5 if log_h > 16 else 40))
        m = rand_field(rng, (1 << log_h, w))
        inv = bool(rng.integers(0, 2))
        assert np.array_equal(ctx.dft_batch(m, inverse=inv), oracle.dft_batch(m, inverse=inv)), ("dft", log_h, w, inv)
    done("ms_dft_batch", n)
    for _ in range(n):
        log_h = int(rng.choice([0, 1, 2, 3, 5, 8, 10, 12, 13, 14, 16, 17]))
        lb = int(rng.integers(1, 4))
        w = int(rng.integers(1, 4 if log_h > 14 else 30))
        m = rand_field(rng, (1 << log_h, w))
        assert np.array_equal(ctx.coset_lde_batch(m, lb), oracle.coset_lde_bitrev(m, lb)), ("lde", log_h, lb, w)
    done("ms_coset_lde_batch", n)
    for _ in range(n):
        log_n, log_q, D = int(rng.integers(0, 14)), int(rng.integers(0, 3)), int(rng.integers(1, 3))
        lb = int(rng.integers(max(log_q, 1), 4))
        q = rand_field(rng, (1 << (log_n + log_q), D))
        want = oracle.lde_from_shifted_coefficients(oracle.shifted_quotient_slices(q, 1 << log_q), lb)
        assert np.array_equal(ctx.quotient_lde(q, log_n, log_q, lb), want), ("quotient_lde", log_n, log_q, D, lb)
    done("ms_quotient_lde", n)
    for _ in range(n):  # Merkle trees: 1-5 matrices of mixed heights (several injection layers), wide rows, caps
        nm = int(rng.integers(1, 6))
        shapes = [(1 << int(rng.integers(0, 13)), int(rng.integers(1, 200 if rng.random() < 0.15 else 20))) for _ in range(nm)]
        cap_h = int(rng.integers(0, 4))
        mats = [rand_field(rng, s) for s in shapes]
        g, o = pkg.Mmcs(ctx, mats, cap_h), oracle.Mmcs(mats, cap_h)
        assert g.cap == o.cap, ("mmcs cap", shapes, cap_h)
        maxh = max(s[0] for s in shapes)
        for index in {0, maxh - 1, int(rng.integers(0, maxh)), int(rng.integers(0, maxh))}:
            gv, gp = g.open(index)
            ov, op = o.open(index)
            assert np.array_equal(gv, ov) and gp == op, ("mmcs open", shapes, cap_h, index)
    done("ms_mmcs_commit / open", n)
    for _ in range(n):  # stage-2 traces: 0-40 lookups with 0-70 arguments each (both fingerprint paths, several batches)
        h = 1 << int(rng.integers(0, 13))
        L = int(rng.integers(1, 41 if rng.random() < 0.2 else 8))
        widths = [int(rng.integers(0, 71 if rng.random() < 0.05 else 7)) for _ in range(L)]
        offs = np.concatenate([[0], np.cumsum(widths)]).astype(np.uint64)
        mult = rand_field(rng, (h, L))
        aw = int(offs[-1])
        args = rand_field(rng, (h, max(aw, 1)))[:, :aw]
        beta = [int(x) for x in rng.integers(2, P, 2, dtype=np.uint64)]
        gamma = [int(x) for x in rng.integers(2, P, 2, dtype=np.uint64)]
        acc = [int(x) for x in rand_field(rng, 2)]
        gt, ga = ctx.stage2_trace(mult, offs, args, beta, gamma, acc)
        ot, oa = oracle.stage2_trace(mult, offs, args, beta, gamma, acc)
        assert ga == oa and np.array_equal(gt, ot), ("stage2", h, widths)
    done("ms_stage2_trace", n)
    fe = pkg.frontend
    for _ in range(n):  # ragged claim sets, on both sides of the host/device switch
        nc = int(rng.choice([0, 1, 5, 200, 300, 5000]))
        claims = [[int(x) for x in rand_field(rng, int(rng.integers(0, 7)))] for _ in range(nc)]
        packed = fe.pack_claims(claims)
        beta = [int(x) for x in rng.integers(2, P, 2, dtype=np.uint64)]
        gamma = [int(x) for x in rng.integers(2, P, 2, dtype=np.uint64)]
        assert ctx.claims_accumulator(packed, beta, gamma) == oracle.claims_accumulator(packed, beta, gamma), ("claims", nc)
    done("ms_claims_accumulator", n)
    for _ in range(n):
        ln = int(rng.choice([0, 1, 63, 64, 65, 1023, 1024, 1025, 4096, 4097, int(rng.integers(0, 3_000_000))]))
        data = rng.integers(0, 256, ln, dtype=np.uint8).tobytes()
        assert ctx.blake3(data) == oracle.hash_bytes(data), ("blake3", ln)
    done("ms_blake3", n)
    print("OK: %d random cases per entry point, every result identical to the oracle's" % n)


if __name__ == "__main__":
    main()
