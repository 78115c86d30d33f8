set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_smp_gpu.py -q -x -m gpu -k "fused_small_launch" 2>&1 | grep -E "passed|failed|Error" | tail -5
