#!/bin/bash
# HBM traffic of the level-7 operator applies (k_apply_slab2) by the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in separate passes, FETCH_SIZE doubled on gfx950 (wide streaming reads), KB -> bytes.  Output: gpurun_out/r05_level7_pmc.txt
set -e
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/l7pmc; rm -rf $O; mkdir -p $O
CMD="python3 $R/bench.py --levels 7 --width 16 --sigma-high 100 --steps 2 --warmup 1 --no-cpu-baseline --no-time-to-tolerance --no-level-report --tune-placement 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $CMD > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $CMD > $O/write.log 2>&1
python3 - <<PY | tee $R/gpurun_out/r05_level7_pmc.txt
import csv, glob, collections
def read(d, name):
    f = glob.glob("$O/" + d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name and "k_apply_slab2" in r["Kernel_Name"]:
            k = r["Kernel_Name"]
            b = k.index("k_apply_slab2")
            acc[k[b:k.index("(", b)]].append(float(r["Counter_Value"]))
    return acc
fe, wr = read("fetch", "FETCH_SIZE"), read("write", "WRITE_SIZE")
dofs = 47905 * 24576
# algorithmic bytes per DOF of the instantiations a V-cycle uses: <FUSED, NS, SRC, OUT, loaders, SLOT>
alg = {"<false, 1, true, true, 4, false>": (16, 8, "residual (x, b in; r out)"),
       "<true, 1, false, true, 4, false>": (8, 8, "CG step 0 (r in; Ap out)"),
       "<true, 2, false, false, 4, false>": (16, 0, "dead last step (r, p in)"),
       "<true, 2, false, true, 4, false>": (16, None, "p-update steps (r, p in; Ap [+ spare] out: 8 or 16 B/DOF written)"),
       "<true, 3, false, true, 4, false>": (24, 24, "full step (r, p, x in; p, x, Ap out)"),
       "<true, 3, true, true, 4, false>": (32, 16, "local residual with two x-updates (x, p, r, b in; x, r out)"),
       "<false, 1, false, true, 8, true>": (8, 8.0 * 6545 / 47905, "restriction through the window (r in; coarse b out)")}
print("level 7, k_apply_slab2: HBM bytes per launch by PMC (FETCH_SIZE x 2 x 1024, WRITE_SIZE x 1024) against algorithmic bytes")
tf = tw = af = aw = 0.0
for k in sorted(fe):
    t = k[len("k_apply_slab2"):]
    f = sum(fe[k]) / len(fe[k]) * 2048.0
    w = sum(wr.get(k, [0.0])) / max(len(wr.get(k, [])), 1) * 1024.0
    a = alg.get(t)
    line = f"{t:38s} launches {len(fe[k]):3d}  read {f / 1e9:7.2f} GB  written {w / 1e9:7.2f} GB"
    if a:
        ar = a[0] * dofs
        line += f"   algorithmic read {ar / 1e9:6.2f} GB (x{f / ar:5.3f})"
        if a[1] is not None:
            aw_ = a[1] * dofs
            line += f", written {aw_ / 1e9:6.2f} GB" + (f" (x{w / aw_:5.3f})" if aw_ > 0 else "")
        line += "   " + a[2]
    print(line)
PY
rm -rf $O
