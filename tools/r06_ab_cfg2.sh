set -u
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do
for lib in "" bsld0 bsst0 bs00; do
  p=""; [ -n "$lib" ] && p=$R/graphflow_amd/csrc/libgf_hip_$lib.so
  GF_HIP_LIBRARY=$p python bench.py --workload cfg2 --steps 50 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('lib=[%s]' % '$lib', d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
