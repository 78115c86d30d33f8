set -u
cd $GRAFT_REPO_ROOT
for cfg in "48 48 10 256" "48 48 32 256"; do
for e in 1 0; do echo "== $cfg GF_SMP_BIG_FIELDS=$e"; GF_SMP_BIG_FIELDS=$e python tools/big_fields_time.py $cfg 2>&1 | grep -v amdgpu.ids | sed -n 2,2p; done; done
timeout 3000 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
