#!/bin/bash
# cfg3 step time and the named kernels' ms per step under a list of environment settings, one line each.
# usage: KERNELS="smpf_vectors smpf_wgrad" tools/cfg3_sweep.sh "GF_X=1" "GF_X=0" ...
for cfg in "$@"; do
  env $cfg python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | KERNELS="$KERNELS" python -c "
import sys, json, os
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['kernel_ms_per_step']
names = os.environ.get('KERNELS', '').split() or list(k)[:8]
print('%-40s %.4f ms  ' % ('$cfg', d['ms_per_step']) + ' '.join('%s=%.3f' % (a.replace('smpf_', ''), k.get(a, 0)) for a in names))
"
done
