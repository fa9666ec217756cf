#!/bin/bash
set -e
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/l4count; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/tools/dev/level4_kernel_count.py > $O/log 2>&1
python3 - <<PY | tee $R/gpurun_out/r05_level4_kernel_count.txt
import csv, glob, collections
f = glob.glob("$O/t/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
fills = [i for i, r in enumerate(rows) if "k_fill_random" in r["Kernel_Name"]]
# the last two k_fill_random launches are the markers
a, b = fills[-2], fills[-1]
seg = rows[a + 1:b]
cnt, tim = collections.Counter(), collections.Counter()
for r in seg:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hmg::", "").replace("(anonymous namespace)::", "")
    n = n.split("<")[0]
    cnt[n] += 1; tim[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
print(f"one V-cycle from level 4 (levels 4, 3, 2 and the level-1 solve; config 3's mesh): {len(seg)} kernel launches, {span:.0f} us from the first start to the last end, {sum(tim.values()):.0f} us of kernel time")
for n, c in cnt.most_common():
    print(f"  {n:28s} {c:4d} launches  {tim[n]:8.1f} us  ({tim[n] / c:6.1f} us each)")
PY
rm -rf $O
