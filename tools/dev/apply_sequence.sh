#!/bin/bash
# Per-launch durations of the finest-level apply launches of the last V-cycles of a bench run (rocprofv3 kernel trace).
# P plain, F fused, C fused with the coarse-grid correction staged in the image, R fused with the restriction in its epilogue.
set -e
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/seq; rm -rf $O; mkdir -p $O
HMG_OPTIONS=${EXTRA_OPTIONS:-} rocprofv3 --kernel-trace --output-format csv -d $O/w1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-time-to-tolerance --no-level-report > $O/w1.log 2>&1
python3 - <<PY | tee $O/sequence.txt
import csv, glob
f = glob.glob("$O/w1/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_apply<3, 1024, 7" in r["Kernel_Name"] or "k_apply<3, 512, 13" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
def tag(n):
    a = n[n.index("<") + 1:n.index(">")].replace(" ", "").split(",")
    fused, cg, rs = a[3] == "true", len(a) > 6 and a[6] == "true", len(a) > 7 and a[7] == "true"
    return "R" if rs else "C" if cg else "F" if fused else "P"
t = [tag(r["Kernel_Name"]) for r in rows]
n = 9
for name, sl in (("last V-cycle", slice(-n, None)), ("previous    ", slice(-2 * n, -n))):
    print(name, " ".join("%s%.2f" % (a, b) for a, b in zip(t[sl], d[sl])), " sum %.2f  mean %.3f" % (sum(d[sl]), sum(d[sl]) / n))
# every launch of the last V-cycle that takes more than a millisecond, in order (applies as above, ru = r-update, xp = x/p-update)
allr = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last = int(rows[-n]["Start_Timestamp"])
short = lambda k: tag(k) if "k_apply<" in k else "ru" if "rupdate" in k else "xp" if "xp_update" in k else "x2" if "x2_update" in k else k.split("(")[0].split("::")[-1][:12]
big = [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in allr if int(r["Start_Timestamp"]) >= last]
big = [(a, b) for a, b in big if b > 1.0]
print("whole last V-cycle, launches > 1 ms:", " ".join("%s%.2f" % ab for ab in big), " sum %.2f" % sum(b for a, b in big))
PY
rm -rf $O/w1
