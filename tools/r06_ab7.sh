set -u
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do for lib in "" w124 w376; do
p=""; [ -n "$lib" ] && p=$R/graphflow_amd/csrc/libgf_hip_$lib.so
GF_HIP_LIBRARY=$p python bench.py --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['roofline']['kernel_ms_per_step']
print('lib=[%-5s] %.3f ms |' % ('$lib', d['ms_per_step']), 'wgrad=%.3f products_bwd=%.3f' % (k.get('smpf_wgrad',0), k.get('smpf_products_bwd',0)))"
done; done
