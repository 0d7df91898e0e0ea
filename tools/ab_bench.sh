#!/bin/bash
# Same-box A/B timing (GPU boxes differ by >10 %): alternate bench runs of the in-tree library and tools/ab/libwrsn_base.so.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-ab}; mkdir -p $OUT
ARGS="--cpu-seconds 0 --steps 60 --warmup 15 --no-blocking-run ${BENCH_ARGS:-}"
for rep in 1 2; do
  for which in base new; do
    if [ $which = base ]; then export WRSN_HIP_LIB=$ROOT/tools/ab/libwrsn_base.so; else unset WRSN_HIP_LIB; fi
    timeout -k 10 200 python $ROOT/bench.py $ARGS > $OUT/${which}_$rep.log 2>&1
    tail -1 $OUT/${which}_$rep.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$which $rep', round(d['value']), d['kernels'])"
  done
done
