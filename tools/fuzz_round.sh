#!/bin/bash
# The round's differential fuzzing on the GPU box: whole path (both fields, long claim lists, wide / tall systems, many circuits,
# wide FRI folds), the joint prover on thread ranks (uniform and general ownership, long claim lists, tall caps), the verifier.
# usage: tools/fuzz_round.sh out.txt [seed base, default 60]
OUT=${1:-gpurun_out/fuzz_round.txt}
S=${2:-60}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
run() { echo "== $*" >> $OUT; env "$@" >> $OUT 2>&1 || echo "FAILED: $*" >> $OUT; }
: > $OUT
run MSAMD_NO_JIT=1 python3 tools/fuzz_parity.py 400 $((S + 1))
run FUZZ_CLAIMS=1 MSAMD_NO_JIT=1 python3 tools/fuzz_parity.py 200 $((S + 2))
run FUZZ_BIG=1 MSAMD_NO_JIT=1 python3 tools/fuzz_parity.py 80 $((S + 3))
run FUZZ_MANY=1 MSAMD_NO_JIT=1 python3 tools/fuzz_parity.py 100 $((S + 4))
run FUZZ_ARITY=1 MSAMD_NO_JIT=1 python3 tools/fuzz_parity.py 150 $((S + 5))
run FUZZ_FIELD=babybear MSAMD_NO_JIT=1 python3 tools/fuzz_parity.py 200 $((S + 6))
run X=1 python3 tools/fuzz_parity.py 40 $((S + 7))
run MSAMD_SHARDED_FUZZ_CASES=150 python3 -m pytest tests/test_gpu_sharded.py -q -m gpu -s -k "random_systems and thread_ranks or general_ownership_random"
run X=1 python3 tools/fuzz_verifier.py 600 $((S + 8))
tail -n 40 $OUT
