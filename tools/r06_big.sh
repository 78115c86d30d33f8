set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_dist_gpu.py -q -x -m gpu -s -k "mock or one_gpu" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -30
