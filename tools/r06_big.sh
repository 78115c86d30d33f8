set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_smp_gpu.py -q -x -m gpu -s -k "fields_above_32" 2>&1 | tail -25
