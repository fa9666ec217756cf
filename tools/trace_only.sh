#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats over the default bench workload -> gpurun_out/<tag>/ (tag = $1)
set -e
T=${1:-trace}
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$T; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/trace.log 2>&1
F=$(ls -t $O/trace/*/*kernel_stats.csv | head -1)
python3 - "$F" > $O/stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("name,calls,total_ms,avg_us,min_us,max_us,pct")
for r in rows:
    print(f"\"{r['Name'][:110]}\",{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.2f},{float(r['MinNs'])/1e3:.2f},{float(r['MaxNs'])/1e3:.2f},{100*float(r['TotalDurationNs'])/tot:.2f}")
PY
rm -rf $O/trace
