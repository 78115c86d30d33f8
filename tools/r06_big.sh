set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_smp_gpu.py -q -x -m gpu -s -k "dense_graphs" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -25
