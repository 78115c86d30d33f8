set -u
R=$GRAFT_REPO_ROOT
cd $R
for wl in cfg2 cfg5 cfg3; do bash tools/profile.sh r06a $wl > gpurun_out/r06a_profile_$wl.log 2>&1; done
cat gpurun_out/r06a_pmc_cfg2_hbm_bytes.txt gpurun_out/r06a_pmc_cfg5_hbm_bytes.txt gpurun_out/r06a_pmc_cfg3_hbm_bytes.txt
