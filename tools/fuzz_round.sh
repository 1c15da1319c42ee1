#!/bin/bash
# The round's differential fuzzing on the GPU box: whole path (both fields, long claim lists, wide / tall systems, many circuits,
# wide FRI folds), the joint prover on thread ranks (uniform and general ownership, long claim lists, tall caps), the verifier.
# usage: tools/fuzz_round.sh out.txt
OUT=${1:-gpurun_out/fuzz_round.txt}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
run() { echo "== $*" >> $OUT; env "$@" >> $OUT 2>&1 || echo "FAILED: $*" >> $OUT; }
: > $OUT
run MSAMD_NO_JIT=1 python3 tools/fuzz_parity.py 400 61
run FUZZ_CLAIMS=1 MSAMD_NO_JIT=1 python3 tools/fuzz_parity.py 200 62
run FUZZ_BIG=1 MSAMD_NO_JIT=1 python3 tools/fuzz_parity.py 80 63
run FUZZ_MANY=1 MSAMD_NO_JIT=1 python3 tools/fuzz_parity.py 100 64
run FUZZ_ARITY=1 MSAMD_NO_JIT=1 python3 tools/fuzz_parity.py 150 65
run FUZZ_FIELD=babybear MSAMD_NO_JIT=1 python3 tools/fuzz_parity.py 200 66
run X=1 python3 tools/fuzz_parity.py 40 67
run MSAMD_SHARDED_FUZZ_CASES=150 python3 -m pytest tests/test_gpu_sharded.py -q -m gpu -s -k "random_systems and thread_ranks or general_ownership_random"
run X=1 python3 tools/fuzz_verifier.py 600 68
tail -n 40 $OUT
