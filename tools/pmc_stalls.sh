#!/bin/bash
# Memory-system stall counters of the V-cycle kernels (dev): separate rocprofv3 --pmc passes, output gpurun_out/pmc_stalls/
set -e
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc_stalls; rm -rf $O; mkdir -p $O
CMD="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum --output-format csv -d $O/p1 -- $CMD > $O/p1.log 2>&1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum --output-format csv -d $O/p2 -- $CMD > $O/p2.log 2>&1
rocprofv3 --pmc TCC_TAG_STALL_sum TCC_IB_STALL_sum GRBM_GUI_ACTIVE TCC_BUSY_sum --output-format csv -d $O/p3 -- $CMD > $O/p3.log 2>&1
echo collected
