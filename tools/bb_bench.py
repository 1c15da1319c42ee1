"""Timing of BASELINE config 4 (BabyBear / Poseidon2, MulAir at 2^LOG rows, the reference's test-suite parameters):
ms per proof and trace rows per second with the witness resident in HBM, per-stage host clocks, and - with --oracle -
the CPU restatement on the same input beside it. Not the bench line (bench.py measures config 2); a parity-test
configuration measured for the record.   usage: python3 tools/bb_bench.py [LOG_ROWS=20] [STEPS=10] [--oracle] [--bench-params]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    log_rows = int(args[0]) if args else 20
    steps = int(args[1]) if len(args) > 1 else 10
    pkg = load_package()
    fe, bb = pkg.frontend, pkg.babybear
    ctx = pkg.Context(0)
    K = fe.poseidon2_constants()
    with fe.field(fe.BABYBEAR):
        params = fe.bench_params() if "--bench-params" in sys.argv else fe.test_params()  # blowup 4, 100 queries, 10 + 10 PoW bits
        g = bb.System.new(ctx, params, fe.mul_air_inputs(), K)
        trace = fe.mul_air_trace(1 << log_rows)
        packed = fe.pack_claims([])
    w = g.witness([trace], packed)
    for _ in range(3):
        proof = g.prove_multiple_claims(w, want_times=True)
    ctx.sync()
    t0 = time.perf_counter()
    stages = {}
    for _ in range(steps):
        p = g.prove_multiple_claims(w, want_times=True)
        for k, v in p.stage_ms.items():
            stages[k] = stages.get(k, 0.0) + v / steps
    ms = (time.perf_counter() - t0) * 1e3 / steps
    assert g.verify(packed, proof.to_bytes()) == 0, "the library's own verifier rejects the proof"
    rec = {"workload": "config 4: MulAir 2^%d rows, BabyBear/Ext4/Poseidon2, %s" % (
        log_rows, "blowup 4, 100 queries, 10 + 10 proof-of-work bits" if "--bench-params" in sys.argv else "blowup 2, 64 queries"), "ms_per_proof": round(ms, 3),
           "trace_rows_per_s": round((1 << log_rows) / ms * 1e3), "proof_bytes": len(proof.to_bytes()),
           "stage_ms": {k: round(v, 3) for k, v in stages.items()}}
    if "--oracle" in sys.argv:
        import oracle_bb as ob
        o = ob.System(g.blob)
        t0 = time.perf_counter()
        want = o.prove([trace], packed)
        dt = time.perf_counter() - t0
        assert want == proof.to_bytes()
        rec["cpu_oracle"] = {"ms_per_proof": round(dt * 1e3, 1), "threads": ob.max_threads(), "identical_bytes": True}
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
