#!/bin/bash
# GPU box: kernel trace + the two HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass) of one bench workload;
# summaries land in gpurun_out/<tag>_<workload>_kernel_stats.txt and gpurun_out/<tag>_pmc_<workload>_hbm_bytes.{txt,json}.
# usage: tools/profile.sh <tag> <cfg2|cfg3|cfg5>      (run from the repo root)
set -u
TAG=${1:-r02}
WL=${2:-cfg3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS="--workload $WL --no-cpu-baseline --no-extra --repeats 1"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_${WL}_trace -o $WL -- python $R/bench.py --steps 20 --warmup 3 $ARGS > $R/gpurun_out/prof_${TAG}_${WL}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_${WL}_fetch -- python $R/bench.py --steps 2 --warmup 1 $ARGS > $R/gpurun_out/prof_${TAG}_${WL}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_${WL}_write -- python $R/bench.py --steps 2 --warmup 1 $ARGS > $R/gpurun_out/prof_${TAG}_${WL}_write.log 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/prof_${TAG}_${WL}_trace/*results.db > gpurun_out/${TAG}_${WL}_kernel_stats.txt
# bench.py runs warm-up + the per-kernel pass + the timed pass: 1 + 2 + 2 = 5 steps per process
python tools/pmc_cfg3.py gpurun_out/prof_${TAG}_${WL}_fetch gpurun_out/prof_${TAG}_${WL}_write 5 gpurun_out/${TAG}_pmc_${WL}_hbm_bytes.json "${GF_COMMIT:-$(cat .gf_commit 2>/dev/null)}" > gpurun_out/${TAG}_pmc_${WL}_hbm_bytes.txt
rm -rf gpurun_out/prof_${TAG}_${WL}_fetch gpurun_out/prof_${TAG}_${WL}_write gpurun_out/prof_${TAG}_${WL}_trace
head -16 gpurun_out/${TAG}_${WL}_kernel_stats.txt
cat gpurun_out/${TAG}_pmc_${WL}_hbm_bytes.txt
