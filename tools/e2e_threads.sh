for t in 4 8 16 32 ""; do
  if [ -n "$t" ]; then export GF_PREP_THREADS=$t; else unset GF_PREP_THREADS; fi
  python bench.py --steps 40 --repeats 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); print('GF_PREP_THREADS=%s step %.3f e2e %.3f ratio %.3f prep_s %.4f' % (os.environ.get('GF_PREP_THREADS','default'), d['ms_per_step'], d['end_to_end']['ms_per_step'], d['end_to_end']['ms_per_step']/d['ms_per_step'], d['config']['prep_s']))"
done
nproc; uptime
