set -u
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2 3; do for lib in "" old; do
p=""; [ -n "$lib" ] && p=$R/graphflow_amd/csrc/libgf_hip_$lib.so
GF_HIP_LIBRARY=$p python bench.py --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['roofline']['kernel_ms_per_step']
print('lib=[$lib] %.3f ms launches %s |' % (d['ms_per_step'], d['roofline']['launches_per_step']), 'diag_bwd=%.3f reduce=%.3f' % (k.get('smpf_diag_gather_bwd',0), k.get('smpf_reduce_pairs',0)))"
done; done
