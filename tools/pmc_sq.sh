#!/bin/bash
# Run on the GPU box: SQ / instruction-cache counter passes of the step-kernel workload (tools/diag_run.py), 8 SQ slots per pass.
# Counters only (no tracing options beside them).  Summaries: gpurun_out/<tag>/pmc_*.csv -> tools/pmc_summarize.py
set -e
TAG=${1:-pmc}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $OUT/p1 -o p --output-format csv -- python3 $ROOT/tools/diag_run.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES SQ_IFETCH -d $OUT/p2 -o p --output-format csv -- python3 $ROOT/tools/diag_run.py > $OUT/p2.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $OUT/p3 -o p --output-format csv -- python3 $ROOT/tools/diag_run.py > $OUT/p3.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU -d $OUT/p4 -o p --output-format csv -- python3 $ROOT/tools/diag_run.py > $OUT/p4.log 2>&1 || true
# lane utilisation of the VALU: thread-cycles against instruction-cycles, and the float64 / transcendental mix
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES -d $OUT/p5 -o p --output-format csv -- python3 $ROOT/tools/diag_run.py > $OUT/p5.log 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_FMA_F32 -d $OUT/p6 -o p --output-format csv -- python3 $ROOT/tools/diag_run.py > $OUT/p6.log 2>&1 || true
python3 $ROOT/tools/pmc_summarize.py $OUT
