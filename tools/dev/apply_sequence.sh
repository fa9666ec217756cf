#!/bin/bash
# Per-launch durations of the finest-level apply launches of one V-cycle, for both workgroup shapes (rocprofv3 kernel trace).
set -e
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/seq; rm -rf $O; mkdir -p $O
for w in 0 1; do
  HMG_OPTIONS=apply_wg512=$w${EXTRA_OPTIONS:+,$EXTRA_OPTIONS} rocprofv3 --kernel-trace --output-format csv -d $O/w$w -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-time-to-tolerance > $O/w$w.log 2>&1
done
python3 - <<PY
import csv, glob
for w in (0, 1):
    f = glob.glob("$O/w%d/*/*kernel_trace.csv" % w)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_apply<3, 1024, 7" in r["Kernel_Name"] or "k_apply<3, 512, 13" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    fused = ["F" if ", true, 6" in r["Kernel_Name"] else "P" for r in rows]
    n = 9
    last = d[-n:]; prev = d[-2 * n:-n]
    print("wg512=%d launches %d  last V-cycle:" % (w, len(d)), " ".join("%s%.2f" % (a, b) for a, b in zip(fused[-n:], last)), " sum %.2f" % sum(last))
    print("            previous V-cycle:", " ".join("%.2f" % b for b in prev), " sum %.2f" % sum(prev))
PY
