set -u
R=$GRAFT_REPO_ROOT
cd $R
GF_MARGINS_OUT=$R/gpurun_out/r05_parity_margins.txt python -m pytest tests -m gpu -q -x > gpurun_out/r05_gpu_suite.txt 2>&1
grep -E "passed|failed" gpurun_out/r05_gpu_suite.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.txt 2>&1; tail -4 gpurun_out/r05_smoke.txt
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
tail -c 300 gpurun_out/r05_bench_default.json
{ for a in "1 32 1024 0" "2 32 1024 0" "2 32 1024 9" "1 10 1024 0" "2 10 1024 9" "1 64 1024 0"; do python tools/physics_time.py $a 2>/dev/null | grep towers; done; } > gpurun_out/r05_physics_times.txt
bash tools/profile.sh r05 cfg3 > gpurun_out/r05_profile_cfg3.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r05_c10 -o c10 -- python $R/bench.py --C 10 --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
cd $R; python tools/rocpd_summary.py gpurun_out/prof_r05_c10/*results.db > gpurun_out/r05_cfg3_C10_kernel_stats.txt; rm -rf gpurun_out/prof_r05_c10
head -12 gpurun_out/r05_cfg3_C10_kernel_stats.txt | cut -c1-60,112-150
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r05_c32 -o c32 -- python $R/bench.py --C 32 --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
cd $R; python tools/rocpd_summary.py gpurun_out/prof_r05_c32/*results.db > gpurun_out/r05_cfg3_C32_kernel_stats.txt; rm -rf gpurun_out/prof_r05_c32
{ python tools/ver67_time.py 10 10 1024; python tools/ver67_time.py 50 10 1024; } 2>/dev/null | grep nContractions > gpurun_out/r05_ver67_times.txt; cat gpurun_out/r05_ver67_times.txt
