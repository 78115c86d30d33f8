set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_smp_gpu.py -q -x -m gpu -s -k "with_fields_above_32" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -25
for nk in 10 50; do for e in 1 0; do GF_SMP_VER6_FUSED=$e GF_SMP_VER7_FUSED=$e python tools/ver67_time.py $nk 10 256 48 2>&1 | tail -1; done; done
