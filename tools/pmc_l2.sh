cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT}
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_l2 -- python $R/bench.py --workload cfg3 --no-cpu-baseline --no-extra --steps 2 --warmup 1 > /dev/null 2>&1
cd $R; python tools/pmc_table.py gpurun_out/pmc_l2 | cut -c1-130 | head -24; rm -rf gpurun_out/pmc_l2
