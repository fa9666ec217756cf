#!/bin/bash
# Per-launch durations of the finest-level apply launches of one V-cycle for two builds: the tree's library and $1 (rocprofv3 kernel trace).
set -e
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/seqab; rm -rf $O; mkdir -p $O
for w in new old new old; do
  if [ $w = old ]; then export HMG_LIB_PATH=$1; else unset HMG_LIB_PATH; fi
  rocprofv3 --kernel-trace --output-format csv -d $O/$w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-time-to-tolerance > $O/$w.log 2>&1
  python3 - <<PY
import csv, glob, os
f = sorted(glob.glob("$O/$w/*/*kernel_trace.csv"), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if "k_apply<3, 1024, 7" in r["Kernel_Name"] or "k_apply<3, 512, 13" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
print("$w:", " ".join("%.2f" % b for b in d[-9:]), " sum %.2f" % sum(d[-9:]), "| prev cycle sum %.2f" % sum(d[-18:-9]))
PY
  rm -rf $O/$w
done
