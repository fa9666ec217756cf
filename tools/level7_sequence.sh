#!/bin/bash
# Per-launch durations, algorithmic bytes and TB/s of the level-7 operator-apply launches of the last V-cycle of a bench run at
# BASELINE config 5's per-GPU share (rocprofv3 kernel trace; 24 576 cells x 47 905 nodes = 1.177e9 DOFs, sigma in {1, 100}).
# $1 = output tag under gpurun_out/, EXTRA_OPTIONS = context options of the run ("apply_slab2=0": k_apply_slab of rounds 3-4).
# Launch order inside hmg_vcycle on the finest level (three CG steps, every exact saving on; csrc/hmg_capi.cpp smooth()):
#   pre-smoother : residual (x, b in; r out: 24 B/DOF) | step 0 (r is p: r in, Ap out: 16) | step 1 (r, p, x in; p, x, Ap out: 48) |
#                  dead last step (r, p in: 16) | local residual with both pending x-updates (x, p, r, b in; x, r out: 48)
#   [restriction through the window: its own pass, k_apply_slab<3,1024,false>: 8 + 1.1]  [prolongation: k_prolong_add_big: 17.1]
#   post-smoother: residual (24) | step 0 (16) | step 1, direction into the spare vector (r, p in; spare, Ap out: 32) |
#                  last step (r, spare in; Ap out: 24)
set -e
T=${1:-l7seq}
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$T; rm -rf $O; mkdir -p $O
HMG_OPTIONS=${EXTRA_OPTIONS:-} rocprofv3 --kernel-trace --output-format csv -d $O/w1 -- python3 $R/bench.py --levels 7 --width 16 --sigma-high 100 --steps 3 --warmup 1 --no-cpu-baseline --no-time-to-tolerance --no-level-report --tune-placement 0 > $O/w1.log 2>&1
python3 - <<PY | tee $O/sequence.txt
import csv, glob
f = glob.glob("$O/w1/*/*kernel_trace.csv")[0]
allr = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
us = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
slab = [r for r in allr if "k_apply_slab" in r["Kernel_Name"]]
dofs = 47905 * 24576
order = [("pre  residual", 24.0), ("pre  step 0", 16.0), ("pre  step 1", 48.0), ("pre  dead last step", 16.0),
         ("local residual + 2 x-updates", 48.0), ("restriction (own pass)", 8.0 + 8.0 * 6545 / 47905), ("post residual", 24.0),
         ("post step 0", 16.0), ("post step 1 (spare vector)", 32.0), ("post last step", 24.0)]
last = slab[-10:]
print("options:", "${EXTRA_OPTIONS:-(defaults)}")
tb = tt = 0.0
for r, (t, b) in zip(last, order):
    gb = b * dofs / 1e9
    if "restriction" not in t:
        tb += gb; tt += us(r)
    k = r["Kernel_Name"]
    print(f"{t:30s} {us(r):8.1f} us  {b:5.1f} B/DOF  {gb:6.2f} GB  {gb / us(r) * 1e3:5.2f} TB/s  {gb / us(r) * 1e3 / 8.0:5.3f} of 8 TB/s   {k[k.index('k_apply'):k.index('(')][:48]}")
print(f"nine apply launches: {tt / 1e3:.2f} ms for {tb:.1f} GB = {tb / tt * 1e3:.2f} TB/s = {tb / tt * 1e3 / 8.0:.3f} of peak")
t0 = int(last[0]["Start_Timestamp"])
rest = {}
for r in allr:
    if int(r["Start_Timestamp"]) >= t0 and "k_apply_slab" not in r["Kernel_Name"]:
        n = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        rest[n] = rest.get(n, 0.0) + us(r)
print("the rest of that V-cycle (ms):", "  ".join(f"{k} {v / 1e3:.2f}" for k, v in sorted(rest.items(), key=lambda kv: -kv[1]) if v > 300.0))
PY
grep '"metric"' $O/w1.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" | tee -a $O/sequence.txt
rm -rf $O/w1
