set -u
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2 3; do
for lib in "" g128; do
  p=""; [ -n "$lib" ] && p=$R/graphflow_amd/csrc/libgf_hip_$lib.so
  GF_HIP_LIBRARY=$p python bench.py --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k=d['roofline']['kernel_ms_per_step']
print('cfg3 lib=[%-5s] %.3f ms |' % ('$lib', d['ms_per_step']), ' '.join('%s=%.3f' % (a.replace('smpf_',''), b) for a, b in list(k.items())[:9]))"
done; done
