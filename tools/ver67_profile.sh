# kernel statistics of the SMP_2D_ver6 / ver7 wirings at C = 10 (op-by-op levels), cfg3 batch.  usage: bash tools/ver67_profile.sh [10|50 ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for k in ${@:-10 50}; do
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v$k -o v$k -- python $R/tools/ver67_time.py $k 10 1024 > $R/gpurun_out/r05_ver67_$k.txt 2>&1
python $R/tools/rocpd_summary.py $R/gpurun_out/prof_v$k/*results.db > $R/gpurun_out/r05_ver67_${k}_kernels.txt
rm -rf $R/gpurun_out/prof_v$k
tail -1 $R/gpurun_out/r05_ver67_$k.txt
head -26 $R/gpurun_out/r05_ver67_${k}_kernels.txt | cut -c1-150
done
