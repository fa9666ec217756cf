#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats and PMC passes over the default bench
# workload.  bash tools/collect_profiles.sh <tag>: output under gpurun_out/prof_<tag>/; tools/summarize_profiles.py <tag>
# turns it into profiles/<tag>_*.
set -e
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; T=${1:-r03}; O=$R/gpurun_out/prof_$T; rm -rf $O; mkdir -p $O
(cd $R && python3 -c "import json, homogenization_jl_amd as h; print(json.dumps(h._lib.fingerprint()))" > $O/fingerprint.json)
CMD="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-time-to-tolerance"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $CMD > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $CMD > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -- $CMD > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pmc_lds -- $CMD > $O/pmc_lds.log 2>&1
tail -1 $O/trace.log | cut -c1-400
echo collected
