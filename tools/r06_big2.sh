set -u
cd $GRAFT_REPO_ROOT
for cfg in "48 48 64 256" "64 64 64 256" "64 64 64 128" "40 40 64 512"; do
for e in 1 0; do echo "== $cfg GF_SMP_BIG_FIELDS=$e"; GF_SMP_BIG_FIELDS=$e python tools/big_fields_time.py $cfg 2>&1 | grep -v amdgpu.ids | head -5; done; done
