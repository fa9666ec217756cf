#!/usr/bin/env python3
"""Per-instantiation means of the rocprofv3 --pmc passes collected by tools/collect_profiles.sh <tag> (gpurun_out/prof_<tag>/)
-> profiles/<tag>_pmc_by_kernel.txt.   python tools/pmc_by_kernel.py r03"""
import collections, csv, glob, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = f"gpurun_out/prof_{tag}"


def latest(d):
    fs = sorted(glob.glob(f"{root}/{d}/*/*counter_collection.csv"), key=os.path.getmtime)
    return fs[-1] if fs else None


def short(n):
    m = re.match(r"void hmg::k_apply<3, 512, 13, (\w+), 6, false, (\w+), (\w+)>", n)
    if m:
        f, cg, rs = m.groups()
        return "k_apply<3,512,13> " + ("plain" if f == "false" else "fused CG" if cg == "true" else "fused RS" if rs == "true" else "fused")
    if n.startswith("hmg::k_cg_rupdate_faces"):
        return "k_cg_rupdate_faces"
    if n.startswith("hmg::k_cg_xp_update"):
        return "k_cg_xp_update"
    m = re.match(r"void hmg::k_apply<3, 256, 4, true, 4, false, false, (\w+)>", n)
    if m:
        return "k_apply<3,256,4> " + ("fused RS" if m.group(1) == "true" else "fused")
    return None


out = collections.OrderedDict()
for d in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write"):
    f = latest(d)
    if not f:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k:
            continue
        if "k_apply" in k and int(r["Grid_Size"]) < 196608 * 256:
            continue
        if "k_cg" in k and int(r["Grid_Size"]) < 100000000:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        for c, vals in v.items():
            out.setdefault(k, {})[c] = (sum(vals) / len(vals), len(vals))
names = sorted({c for v in out.values() for c in v})
with open(f"profiles/{tag}_pmc_by_kernel.txt", "w") as fo:
    fo.write(f"# rocprofv3 --pmc passes of tools/collect_profiles.sh {tag} (final build), mean per launch, full-grid launches of levels 6 and 5 only\n")
    fo.write("# FETCH_SIZE / WRITE_SIZE in KB as reported (FETCH x2 on gfx950, MI355X_MICROARCH.md); SQ_* summed over the chip\n")
    for k, v in out.items():
        fo.write(f"\n{k}   ({max(n for _, n in v.values())} launches)\n")
        for c in names:
            if c in v:
                fo.write(f"    {c:26s} {v[c][0]:18.0f}\n")
        g = lambda c: v[c][0] if c in v else 0.0
        if g("FETCH_SIZE") and g("WRITE_SIZE"):
            fo.write(f"    HBM bytes (FETCH x2 + WRITE) {(2 * g('FETCH_SIZE') + g('WRITE_SIZE')) * 1024 / 1e9:10.2f} GB\n")
        if g("SQ_LDS_IDX_ACTIVE"):
            fo.write(f"    LDS bank-conflict cycles / LDS-active cycles {g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE'):6.2f}\n")
        if g("SQ_WAVE_CYCLES"):
            fo.write(f"    wave cycles waiting for an instruction to return {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):6.2f}\n")
print(open(f"profiles/{tag}_pmc_by_kernel.txt").read()[:600])
