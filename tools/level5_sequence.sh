#!/bin/bash
# Per-launch durations, algorithmic bytes and TB/s of the level-5 apply launches of the last V-cycle of a bench run
# (rocprofv3 kernel trace; config 3: 196 608 cells x 969 nodes = 1.905e8 level-5 DOFs).  $1 = output tag under gpurun_out/,
# EXTRA_OPTIONS = context options of the run ("apply_wave=0": the 256-thread workgroup kernel of rounds 2-3).
# Launch order inside hmg_vcycle on level 5 (two CG steps, lazy dead tail, restriction in the epilogue, folded correction):
#   F step 0 (r is p: 16 B/DOF) | F dead step (reads r, p: 16) | R local residual, two pending x-updates, x = 0 not read,
#   restriction in the epilogue (b, r, p in; x, coarse b out: 32 + 1.4) | C residual with the coarse-grid correction
#   (x, coarse x, b in; x, r out: 32 + 1.4) | F step 0 (16) | F last step (option lazy_post: reads r, p: 16; LAST_B=40 for runs
#   with lazy_post=0 or builds before round 4, where it also reads x and writes p, x)
set -e
T=${1:-l5seq}
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$T; rm -rf $O; mkdir -p $O
HMG_OPTIONS=${EXTRA_OPTIONS:-} rocprofv3 --kernel-trace --output-format csv -d $O/w1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-time-to-tolerance --no-level-report --tune-placement 0 > $O/w1.log 2>&1
python3 - <<PY | tee $O/sequence.txt
import csv, glob
f = glob.glob("$O/w1/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "k_apply_wave" in r["Kernel_Name"] or "k_apply<3, 256, 4" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dofs = 969 * 196608
def tag(n):
    a = n[n.index("<") + 1:n.index(">")].replace(" ", "").split(",")
    if "k_apply_wave" in n:
        fused, cg, rs = a[0] == "true", a[1] == "true", a[2] == "true"
    else:
        fused, cg, rs = a[3] == "true", len(a) > 6 and a[6] == "true", len(a) > 7 and a[7] == "true"
    return "R" if rs else "C" if cg else "F" if fused else "P"
n = 6
last = rows[-n:]
print("kernel:", last[0]["Kernel_Name"][:60])
tr = 32.0 + 8.0 * 165 / 969
order = [("F step 0", 16.0), ("F dead step", 16.0), ("R local residual + restriction", tr), ("C correction residual", tr),
         ("F step 0", 16.0), ("F last step", float("${LAST_B:-16}"))]
tot_b = tot_t = 0.0
for r, (t, b) in zip(last, order):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    bpd = {t: b}
    gb = bpd[t] * dofs / 1e9
    tot_b += gb; tot_t += us
    print(f"{t:32s} {us:8.1f} us  {bpd[t]:5.1f} B/DOF  {gb:6.2f} GB  {gb / us * 1e3:5.2f} TB/s  {gb / us * 1e3 / 8.0:5.3f} of 8 TB/s")
print(f"six launches: {tot_t / 1e3:.3f} ms for {tot_b:.1f} GB = {tot_b / tot_t * 1e3:.2f} TB/s = {tot_b / tot_t * 1e3 / 8.0:.3f} of peak")
PY
grep '"metric"' $O/w1.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])" | tee -a $O/sequence.txt
rm -rf $O/w1
