#!/bin/bash
# cfg5 (RisiContraction_50) step time and per-kernel times under a list of environment settings, one line each.
# usage: tools/cfg5_sweep.sh "GF_X=1 GF_Y=2" "GF_X=0" ...
for cfg in "$@"; do
  env $cfg python bench.py --workload cfg5 --steps 50 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['kernel_ms']
print('%-44s %.4f ms  ' % ('$cfg', d['ms_per_step']) + ' '.join('%s=%.3f' % (a.replace('fam_', ''), b) for a, b in k.items()))
"
done
