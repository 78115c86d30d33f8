set -u
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
python bench.py --steps 50 --warmup 5 --repeats 3 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('cfg3', d['ms_per_step'], d.get('end_to_end'))"
GF_FORCE_DIST=1 python bench.py --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline --no-extra 2>&1 | tail -c 600
