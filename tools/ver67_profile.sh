R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for k in 10 50; do
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v$k -o v$k -- python $R/tools/ver67_time.py $k 10 1024 > $R/gpurun_out/r05_ver67_$k.txt 2>&1
python $R/tools/rocpd_summary.py $R/gpurun_out/prof_v$k/*results.db > $R/gpurun_out/r05_ver67_${k}_kernels.txt
rm -rf $R/gpurun_out/prof_v$k
done
cd $R
python tools/ver67_time.py 10 10 1024; python tools/ver67_time.py 50 10 1024
head -24 gpurun_out/r05_ver67_10_kernels.txt | cut -c1-160
head -24 gpurun_out/r05_ver67_50_kernels.txt | cut -c1-160
