#!/bin/bash
# Build a variant of libgf_hip.so with extra compiler flags (here, no GPU needed) for an A/B run on the GPU box:
#   tools/ab_lib.sh <name> "<flags>" [sources to rebuild ...]   ->  graphflow_amd/csrc/libgf_hip_<name>.so   (select with GF_HIP_LIBRARY=<path>)
set -e
NAME=$1; FLAGS=$2; shift 2
cd "$(dirname "$0")/../graphflow_amd/csrc"
OBJS=""
for f in gf_capi contract18 contract_families mixers smp smp_fused smp_level_c64 smp_level_c64_split smp_level_c64_fwd smp_level_ops gf_dist head smp_model; do
  o=$f.o
  for r in "$@"; do
    if [ "$r" = "$f" ]; then o=/tmp/ab_${NAME}_$f.o; /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include $FLAGS -c $f.hip -o $o; fi
  done
  OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgf_hip_$NAME.so $OBJS smp_prep.o -ldl -lpthread
ls -la libgf_hip_$NAME.so
