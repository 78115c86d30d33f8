#!/bin/bash
# GPU box: memory-path counters of one bench workload's kernels (TLB, L1->L2 requests and latency, L2 hits, EA request sizes, TA stalls),
# one --pmc pass per group (kernel trace only).  usage: [WL=cfg3] tools/pmc_mem.sh <tag>
set -u
TAG=${1:-mem}
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS="--workload ${WL:-cfg3} --no-cpu-baseline --no-extra --steps 2 --warmup 1"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCC_BUSY_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python $R/bench.py $ARGS > /dev/null 2>&1
  (cd $R; python tools/pmc_table.py gpurun_out/pmc_${TAG}_$i | cut -c1-150 | head -16; rm -rf gpurun_out/pmc_${TAG}_$i)
done
