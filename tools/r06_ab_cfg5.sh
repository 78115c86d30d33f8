set -u
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do
for lib in "" fam0 fam1 fam2 fam4; do
  p=""; [ -n "$lib" ] && p=$R/graphflow_amd/csrc/libgf_hip_$lib.so
  GF_HIP_LIBRARY=$p python bench.py --workload cfg5 --steps 50 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('lib=[%-5s]' % '$lib', d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
python -m pytest tests/test_families_gpu.py -q -x -m gpu 2>&1 | tail -2
GF_HIP_LIBRARY= python bench.py --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('cfg3 default', d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
