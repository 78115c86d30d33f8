set -u
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do for lib in "" fam7; do for K in 4 10; do
p=""; [ -n "$lib" ] && p=$R/graphflow_amd/csrc/libgf_hip_$lib.so
echo -n "lib=[$lib] "; GF_HIP_LIBRARY=$p python tools/fam10_time.py $K 2>&1 | tail -1
done; done; done
python -m pytest tests/test_families_gpu.py -q -x -m gpu 2>&1 | grep -E "passed|failed"
