#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 passes of the default bench.py command, summaries into gpurun_out/prof_<tag>/.
#   1. --kernel-trace --stats      -> per-kernel average duration
#   2. --pmc FETCH_SIZE            -> HBM read requests      (separate pass, MI355X_MICROARCH.md HBM section)
#   3. --pmc WRITE_SIZE            -> HBM write requests     (separate pass)
# The program itself follows `--` (no env / bash -c hop).
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 40 --warmup 10 --cpu-seconds 0 --kernel-steps 5 --no-blocking-run ${BENCH_ARGS:-}"   # BENCH_ARGS / SUMMARY_ARGS: another configuration, e.g. "--nodes 1000 --targets 1000 --mcs 8" / "nodes=1000 targets=1000 chargers=8"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o s --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o f --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o w --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/write.log 2>&1
python3 $ROOT/tools/profile_summarize.py $OUT $TAG ${SUMMARY_ARGS:-}
