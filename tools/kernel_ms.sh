# per-kernel ms of a cfg3 step at a channel count.  usage: bash tools/kernel_ms.sh C [C ...]
for c in "$@"; do
python bench.py --workload cfg3 --C $c --steps 50 --warmup 5 --no-cpu-baseline --no-extra --repeats 3 2>/dev/null | tail -1 > gpurun_out/kernel_ms_$c.json
python - $c <<'PY'
import json,sys
d=json.loads(open('gpurun_out/kernel_ms_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print('C', sys.argv[1], d['ms_per_step'], d['timed_region']['ms_per_step_min'], d['timed_region']['ms_per_step_max'])
km=d['roofline']['kernel_ms_per_step']
print('  '.join('%s %.3f'%(k.replace('smpf_',''),v) for k,v in sorted(km.items(), key=lambda x:-x[1])[:16]))
PY
done
