#!/bin/bash
# Run on the GPU box: kernel trace of a short bench run (both streams of a budgeted step call), per-kernel statistics and the
# timeline of three consecutive calls.   tools/trace_split.sh <tag> [lib.so]
TAG=${1:-trace}; LIB=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
if [ -n "$LIB" ]; then export WRSN_HIP_LIB=$ROOT/$LIB; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/kt -o k --output-format csv -- python3 $ROOT/bench.py --steps 30 --warmup 10 --cpu-seconds 0 --kernel-steps 2 --no-blocking-run --min-seconds 0 > $OUT/kt.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "wrsn_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
import collections
agg = collections.defaultdict(list)
for r in rows: agg[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items(): print("%-60s calls %4d avg %8.1f us  min %8.1f max %8.1f" % (k[:60], len(v), sum(v) / len(v), min(v), max(v)))
# timeline of three calls in the middle of the timed loop
obs = [i for i, r in enumerate(rows) if "wrsn_obs_kernel" in r["Kernel_Name"]]
a, b = obs[len(obs) // 2], obs[len(obs) // 2 + 3]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    print("%9.1f -> %9.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Kernel_Name"].split("(")[0][:50]))
PY
