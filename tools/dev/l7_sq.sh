#!/bin/bash
# SQ counters of the level-7 operator applies (k_apply_slab2), two passes of eight.  Output: gpurun_out/r05_level7_sq.txt
# (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave, MI355X_MICROARCH.md)
set -e
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/l7sq; rm -rf $O; mkdir -p $O
CMD="python3 $R/bench.py --levels 7 --width 16 --sigma-high 100 --steps 2 --warmup 1 --no-cpu-baseline --no-time-to-tolerance --no-level-report --tune-placement 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/a -- $CMD > $O/a.log 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/b -- $CMD > $O/b.log 2>&1
python3 - <<PY | tee $R/gpurun_out/r05_level7_sq.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in "ab":
    for f in glob.glob("$O/" + d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_apply_slab2" not in k: continue
            b = k.index("k_apply_slab2")
            acc[k[b:k.index("(", b)]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
         "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD",
         "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU"]
print("level 7, k_apply_slab2, SQ counters per launch (mean over launches); dofs per launch = 47905 x 24576 = 1.177e9")
for k in sorted(acc):
    print(k)
    v = {n: (sum(acc[k][n]) / len(acc[k][n]) if acc[k][n] else float("nan")) for n in names}
    for n in names:
        extra = ""
        if n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
            extra = f"   {v[n] / v['SQ_WAVE_CYCLES']:.3f} of wave cycles"
        if n == "SQ_LDS_BANK_CONFLICT":
            extra = f"   {v[n] / v['SQ_LDS_IDX_ACTIVE']:.3f} of LDS-array cycles"
        if n.startswith("SQ_INSTS"):
            extra = f"   {v[n] * 64 / (47905 * 24576):.2f} lane-instructions per DOF"
        print(f"   {n:24s} {v[n]:16.0f}{extra}")
PY
rm -rf $O
