#!/usr/bin/env python3
"""Condenses gpurun_out/prof_r01 (tools/collect_profiles.sh) into the tracked files under profiles/."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    return f[-1] if f else None


stats = one("trace/*/*kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(dst, f"{tag}_vcycle_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-time-to-tolerance (1 + 2 V-cycles, 2 with HIP events around the finest-level applies, 2 with every level timed, the level shares, the placement tuner + 6 V-cycles)\n")
    f.write("name,calls,total_ms,avg_us,min_us,max_us,pct\n")
    for r in rows:
        f.write(f"\"{r['Name']}\",{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.2f},"
                f"{float(r['MinNs'])/1e3:.2f},{float(r['MaxNs'])/1e3:.2f},{100*float(r['TotalDurationNs'])/tot:.2f}\n")


def counters(pattern):
    f = one(pattern)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f:
        return acc
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_apply<3, 1024, 7" in n or "k_apply<3, 512, 13" in n:       # the finest level of config 3 (either workgroup shape)
            acc["k_apply_L6_fused" if "true" in n else "k_apply_L6_plain"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        elif "k_apply_wave<" in n or "k_apply<3, 256, 4" in n:             # level 5: one wave per cell (round 4) / the workgroup kernel
            acc["k_apply_L5"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        elif "k_cg_rupdate" in n:
            acc["k_cg_rupdate"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


out = {}
for pat in ("pmc_fetch/*/*counter_collection.csv", "pmc_write/*/*counter_collection.csv",
            "pmc_sq/*/*counter_collection.csv", "pmc_lds/*/*counter_collection.csv"):
    for k, v in counters(pat).items():
        for c, vals in v.items():
            if k == "k_cg_rupdate":
                vals = sorted(vals)[-max(1, len(vals) // 4):]      # finest-level launches only
            out.setdefault(k, {})[c] = {"launches": len(vals), "mean": sum(vals) / len(vals)}
with open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w") as f:
    json.dump(out, f, indent=1)

# dominant kernel traffic: all finest-level apply launches (plain + fused), per launch
fetch = [], []
f_all, w_all = [], []
for k in ("k_apply_L6_plain", "k_apply_L6_fused"):
    if k in out and "FETCH_SIZE" in out[k]:
        f_all.append((out[k]["FETCH_SIZE"]["mean"], out[k]["FETCH_SIZE"]["launches"]))
        w_all.append((out[k]["WRITE_SIZE"]["mean"], out[k]["WRITE_SIZE"]["launches"]))
if f_all:
    nf = sum(n for _, n in f_all)
    fetch_kb = sum(m * n for m, n in f_all) / nf
    write_kb = sum(m * n for m, n in w_all) / sum(n for _, n in w_all)
    rec = {"hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
           "fetch_size_kb_raw_mean": fetch_kb, "write_size_kb_mean": write_kb,
           "correction": "FETCH_SIZE x2 (gfx950 wide streaming reads, MI355X_MICROARCH.md HBM section), KB -> x1024",
           "source": f"profiles/{tag}_pmc_summary.json: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over "
                     "bench.py --steps 2 --warmup 1, mean over the finest-level k_apply launches"}
    # the build the counters were collected on (tools/collect_profiles.sh writes it on the GPU box): bench.py reports the
    # traffic figure only when the build it runs is this one
    fp = os.path.join(src, "fingerprint.json")
    if os.path.exists(fp):
        rec.update(json.load(open(fp)))
    json.dump(rec, open(os.path.join(dst, "apply_traffic.json"), "w"), indent=1)
print(open(os.path.join(dst, f"{tag}_pmc_summary.json")).read()[:3000])
