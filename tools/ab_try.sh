#!/bin/bash
# Same-box timing of candidate libraries: tools/ab_try.sh <tag> libA.so libB.so ... (in-tree library = "tree")
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-try}; mkdir -p $OUT; shift
ARGS="--cpu-seconds 0 --steps 60 --warmup 20 --no-blocking-run ${BENCH_ARGS:-}"
for rep in 1 2; do
  for lib in "$@"; do
    if [ $lib = tree ]; then unset WRSN_HIP_LIB; else export WRSN_HIP_LIB=$ROOT/$lib; fi
    timeout -k 10 200 python $ROOT/bench.py $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $rep', round(d['value']), d['kernels'])"
  done
done
