#!/bin/bash
# GPU box: kernel trace + the two HBM-traffic PMC passes of the default (cfg3) workload; summaries into gpurun_out/<tag>_*.
# usage: tools/profile_cfg3.sh <tag>      (run from the repo root)
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_trace -o cfg3 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $R/gpurun_out/prof_${TAG}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_fetch -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $R/gpurun_out/prof_${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/prof_${TAG}_write -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $R/gpurun_out/prof_${TAG}_write.log 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/prof_${TAG}_trace/*results.db > gpurun_out/${TAG}_cfg3_kernel_stats.txt
python tools/pmc_cfg3.py gpurun_out/prof_${TAG}_fetch gpurun_out/prof_${TAG}_write 5 gpurun_out/${TAG}_pmc_cfg3_hbm_bytes.json > gpurun_out/${TAG}_pmc_cfg3_hbm_bytes.txt
rm -rf gpurun_out/prof_${TAG}_fetch gpurun_out/prof_${TAG}_write gpurun_out/prof_${TAG}_trace
head -30 gpurun_out/${TAG}_cfg3_kernel_stats.txt
cat gpurun_out/${TAG}_pmc_cfg3_hbm_bytes.txt
