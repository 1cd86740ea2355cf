#!/usr/bin/env python3
"""Per-kernel matrix-core counters of a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32
GRBM_GUI_ACTIVE) joined with the per-kernel times of the --kernel-trace --stats pass of the same command.
MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)   (the definition of profiles/r02_prefill_mfma_counters.md)
    python tools/mfma_counters.py <pmc dir> <trace dir> <label> >> profiles/r03_prefill_mfma_counters.md"""
import csv, glob, os, re, sys
from collections import defaultdict

pmc_dir, trace_dir, label = sys.argv[1], sys.argv[2], sys.argv[3]


def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)


acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for f in glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), k)
        if key not in seen:
            seen.add(key); cnt[k] += 1
dur = {}
for f in glob.glob(os.path.join(trace_dir, "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
print("\n### %s\n" % label)
print("| kernel | launches | avg us (trace pass) | %% of GPU time | MFMA-busy | MOPS_F32 per launch |")
print("|---|---|---|---|---|---|")
rows = []
for k, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    frac = busy / (gui / 8.0 * 1024.0) if gui else 0.0
    d = dur.get(k, (cnt[k], 0.0, 0.0))
    rows.append((d[2], k, d[0], d[1], frac, c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0) / max(cnt[k], 1)))
for pct, k, n, us, frac, mops in sorted(rows, reverse=True)[:12]:
    print("| %s | %d | %.1f | %.1f | %.3f | %.3e |" % (k, n, us, pct, frac, mops))
