# GPU box: everything profiles/ holds for a round, in one call:  bash tools/collect_round.sh r06
set -u
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
cd $R
GF_MARGINS_OUT=$R/gpurun_out/${TAG}_parity_margins.txt python -m pytest tests -m gpu -q -x -s > gpurun_out/${TAG}_gpu_suite.txt 2>&1
grep -E "passed|failed" gpurun_out/${TAG}_gpu_suite.txt | tail -2
grep -E "cfg3 in-batch|cfg4 workload" gpurun_out/${TAG}_gpu_suite.txt >> gpurun_out/${TAG}_parity_margins.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.txt 2>&1; tail -4 gpurun_out/${TAG}_smoke.txt
python bench.py > gpurun_out/${TAG}_cfg3_bench.json 2> gpurun_out/${TAG}_bench_default.err
tail -c 300 gpurun_out/${TAG}_cfg3_bench.json
for wl in cfg3 cfg2 cfg5; do bash tools/profile.sh $TAG $wl > gpurun_out/${TAG}_profile_$wl.log 2>&1; done
bash tools/pmc_sq.sh $TAG > gpurun_out/${TAG}_cfg3_sq_counters.txt 2>&1
{ for a in "1 32 1024 0" "2 32 1024 0" "2 32 1024 9" "1 10 1024 0" "2 10 1024 9" "1 64 1024 0"; do python tools/physics_time.py $a 2>/dev/null | grep towers; done; } > gpurun_out/${TAG}_physics_times.txt
tools/micro/copy_probe > gpurun_out/${TAG}_copy_probe.txt 2>&1
ls gpurun_out | grep $TAG
