#!/bin/bash
# cfg2 (RisiContraction_18) step time and per-kernel times under a list of environment settings, one line each.
for cfg in "$@"; do
  env $cfg python bench.py --workload cfg2 --steps 50 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['roofline']['kernel_ms']
print('%-32s %.4f ms  ' % ('$cfg', d['ms_per_step']) + ' '.join('%s=%.3f' % (a.replace('r18_', ''), b) for a, b in k.items()))
"
done
