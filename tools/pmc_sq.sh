#!/bin/bash
# GPU box: SQ counters of the cfg3 step's kernels (issue / wait / LDS): tools/pmc_sq.sh <tag> [kernel filter]
set -u
TAG=${1:-sq}
FLT=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS="--workload ${WL:-cfg3} ${BENCH_EXTRA:-} --no-cpu-baseline --no-extra --steps 2 --warmup 1"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_a -- python $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_b -- python $R/bench.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_c -- python $R/bench.py $ARGS > /dev/null 2>&1
cd $R
for p in a b c; do python tools/pmc_table.py gpurun_out/pmc_${TAG}_$p "$FLT" | cut -c1-160; rm -rf gpurun_out/pmc_${TAG}_$p; done
