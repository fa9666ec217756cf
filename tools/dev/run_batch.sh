R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
export HMG_REHEARSE_WORLD=8
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r8trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-time-to-tolerance --no-level-report > $R/gpurun_out/r8trace.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/r8trace/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def short(k):
    k = k.split("(")[0].replace("hmg::", "").replace("void ", "")
    return k[:60]
# last V-cycle: from the last plain residual of level 6 (k_apply<3, 512, 13, false)
idx = [i for i, r in enumerate(rows) if "k_apply<3, 512, 13, false" in r["Kernel_Name"]]
# the overlapped form launches two parts: find the start of the last V-cycle = the first of the last group
last = idx[-1]
while last - 1 in idx or (last - 2 in idx): last = last - 1 if last - 1 in idx else last - 2
t0 = int(rows[last]["Start_Timestamp"]); tend = int(rows[-1]["End_Timestamp"])
agg = collections.OrderedDict()
for r in rows[last:]:
    k = short(r["Kernel_Name"]); d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += d
print("last V-cycle: %.2f ms wall, %.2f ms in kernels, %d launches" % ((tend - t0) / 1e6, sum(a[1] for a in agg.values()), sum(a[0] for a in agg.values())))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]: print("%8.3f ms %5d  %s" % (a[1], a[0], k))
print("sequence of launches > 0.5 ms:")
print(" ".join("%s:%.2f" % (short(r["Kernel_Name"])[:22], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows[last:] if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 5e5))
PY
rm -rf $R/gpurun_out/r8trace
