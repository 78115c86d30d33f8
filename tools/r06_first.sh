set -u
R=$GRAFT_REPO_ROOT
cd $R
GF_MARGINS_OUT=$R/gpurun_out/r06_parity_margins_start.txt timeout 3000 python -m pytest tests -m gpu -q -x -s > gpurun_out/r06_gpu_suite_start.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r06_gpu_suite_start.txt | tail -3
grep -E "cfg3 in-batch|cfg4 workload" gpurun_out/r06_gpu_suite_start.txt
timeout 900 python bench.py > gpurun_out/r06_bench_start.json 2> gpurun_out/r06_bench_start.err
tail -c 600 gpurun_out/r06_bench_start.json
