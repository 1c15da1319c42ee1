"""Mutation fuzzing of the verifier (ms_verify): valid proofs of several systems are corrupted - bit flips, byte
overwrites, length fields blown up, truncation, extension, spliced chunks - and given to the library's verifier and to the
oracle's. Neither may crash, and both must agree on accept / reject (a corrupted proof that both still accept would be
reported too: it must not happen for these proofs).

usage: python3 tools/fuzz_verifier.py [N_MUTATIONS_PER_PROOF] [SEED]
The oracle is used only as the checker."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402


def _only_unread_fields_differ(proof: bytes, mutated: bytes, params) -> bool:
    """both verifiers accepted a mutated proof: fine only if nothing they read changed - the proof-of-work witnesses of a phase whose
    bit count is zero (the reference's check_witness returns before looking at them)"""
    import proof_codec

    try:
        a, b = proof_codec.parse(proof), proof_codec.parse(mutated)
    except Exception:
        return False
    for p in (a, b):
        if params.commit_proof_of_work_bits == 0:
            p["opening_proof"]["commit_pow_witnesses"] = None
        if params.query_proof_of_work_bits == 0:
            p["opening_proof"]["query_pow_witness"] = None
    return a == b


def mutate(rng, proof: bytes) -> bytes:
    b = bytearray(proof)
    n = len(b)
    kind = int(rng.integers(0, 8))
    if kind == 0:  # single bit
        b[int(rng.integers(0, n))] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:  # random byte
        b[int(rng.integers(0, n))] = int(rng.integers(0, 256))
    elif kind == 2:  # an aligned u64 replaced by a huge / small value (often a length field or a field element)
        pos = int(rng.integers(0, max(1, n - 8)))
        vals = [0, 1, 2, 0xFFFFFFFF, 0xFFFFFFFF00000001, 0xFFFFFFFFFFFFFFFF, 1 << 40, n, n * 8]
        val = vals[int(rng.integers(0, len(vals)))]
        b[pos:pos + 8] = val.to_bytes(8, "little")
    elif kind == 3:  # truncate
        b = b[: int(rng.integers(0, n))]
    elif kind == 4:  # extend
        b += bytes(int(x) for x in rng.integers(0, 256, int(rng.integers(1, 64))))
    elif kind == 5:  # copy a chunk over another place
        ln = int(rng.integers(1, 128))
        src, dst = int(rng.integers(0, max(1, n - ln))), int(rng.integers(0, max(1, n - ln)))
        b[dst:dst + ln] = b[src:src + ln]
    elif kind == 6:  # zero a run
        ln = int(rng.integers(1, 64))
        pos = int(rng.integers(0, max(1, n - ln)))
        b[pos:pos + ln] = bytes(ln)
    else:  # several bit flips
        for _ in range(int(rng.integers(2, 6))):
            b[int(rng.integers(0, n))] ^= 1 << int(rng.integers(0, 8))
    return bytes(b)


def main():
    n_mut = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pkg = load_package()
    fe = pkg.frontend
    import oracle

    ctx = pkg.Context(0)
    rng = np.random.default_rng(seed)
    bases = []
    t, c = fe.u32_add_bench_witness(1 << 7)
    bases.append(("u32_add bench params", fe.u32_add_system_inputs(), fe.bench_params(), t, c))
    bases.append(("even/odd lookups", fe.even_odd_inputs(), fe.test_params(), fe.even_odd_traces(), [[0, 4, 1]]))
    bases.append(("squares", fe.squares_inputs(), fe.test_params(), fe.squares_traces(16), []))
    bases.append(("pythagorean", fe.pythagorean_inputs(), fe.test_params(), [fe.pythagorean_trace(64)], []))
    bases.append(("u32_add, caps + 4-coefficient final polynomial", fe.u32_add_system_inputs(),
                  fe.Params(log_blowup=2, cap_height=2, log_final_poly_len=2, num_queries=10, commit_proof_of_work_bits=3,
                            query_proof_of_work_bits=4), t, c))
    # FRI rounds of arity 8 / 4 (max_log_arity 3 / 2): the openings carry log_arity and 2^a - 1 sibling values per round
    bases.append(("u32_add, FRI arity 8, proof of work", fe.u32_add_system_inputs(),
                  fe.Params(log_blowup=2, max_log_arity=3, num_queries=12, commit_proof_of_work_bits=3, query_proof_of_work_bits=4), t, c))
    bases.append(("u32_add, FRI arity 4, caps, 2-coefficient final polynomial", fe.u32_add_system_inputs(),
                  fe.Params(log_blowup=1, cap_height=1, log_final_poly_len=1, max_log_arity=2, num_queries=10, commit_proof_of_work_bits=2,
                            query_proof_of_work_bits=2), t, c))
    t0 = time.time()
    total = same_bytes = unread = 0
    for name, inputs, params, traces, claims in bases:
        g = pkg.System.new(ctx, params, inputs)
        o = oracle.System(g.blob)
        packed = fe.pack_claims(claims)
        proof = g.prove_multiple_claims(g.witness(traces, packed)).to_bytes()
        assert g.verify(packed, proof) == 0 and o.verify(packed, proof) == 0, name
        rejected = 0
        for k in range(n_mut):
            m = mutate(rng, proof)
            if m == proof:
                same_bytes += 1
                continue
            a, b = g.verify(packed, m), o.verify(packed, m)
            assert (a == 0) == (b == 0), "%s, mutation %d: library verdict %d, oracle verdict %d" % (name, k, a, b)
            if a == 0 and _only_unread_fields_differ(proof, m, params):
                unread += 1  # a proof-of-work witness at zero bits is not read (DeterministicPow, src/types.rs:75-80): not a corruption
                continue
            assert a != 0, "%s, mutation %d: a corrupted proof (%d bytes) was accepted by both verifiers" % (name, k, len(m))
            rejected += 1
            total += 1
        print("[fuzz %6.1fs] %-50s %d corrupted proofs rejected by both verifiers" % (time.time() - t0, name, rejected), flush=True)
    print("OK: %d corrupted proofs, no crash, no disagreement, none accepted (%d mutations were no-ops, %d touched only a proof-of-work "
          "witness that zero bits leave unread)" % (total, same_bytes, unread))


if __name__ == "__main__":
    main()
