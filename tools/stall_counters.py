#!/usr/bin/env python3
"""Where a kernel's wave cycles go: joins the rocprofv3 --pmc passes of tools/stall_counters.sh (SQ_* issue / wait buckets, instruction mixes, vector-L1 and L2 hits).
WAIT_ANY (parked at s_waitcnt / barrier) + WAIT_INST_ANY (issue stall: matrix-core RAW, pipe busy) + ACTIVE_INST_ANY ~ WAVE_CYCLES (MI355X_MICROARCH.md, PMC slots).
    python tools/stall_counters.py <dir with pass_*/> <label>"""
import csv, glob, os, re, sys
from collections import defaultdict

root, label = sys.argv[1], sys.argv[2]


def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)


acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(root, "pass_*", "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), k, r["Counter_Name"])
        if key not in seen:
            seen.add(key); cnt[k][r["Counter_Name"]] += 1
print("\n### %s\n" % label)
names = sorted({c for k in acc for c in acc[k]})
print("counters collected: " + " ".join(names) + "\n")
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0.0))[:8]:
    c = acc[k]
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if not wc:
        continue
    n = max(cnt[k].get("SQ_WAVE_CYCLES", 1), 1)
    print("* `%s` (%d launches): wave quad-cycles per launch %.3e" % (k, n, wc / n))
    for name in names:
        if name == "SQ_WAVE_CYCLES":
            continue
        v = c.get(name, 0.0) / max(cnt[k].get(name, 1), 1)
        extra = ""
        if name.startswith("SQ_WAIT") or name.startswith("SQ_ACTIVE_INST") or name in ("SQ_INST_CYCLES_VMEM", "SQ_INST_CYCLES_SMEM"):
            extra = " = %.3f of the wave cycles" % (v / (wc / n))
        print("  * %s %.4e%s" % (name, v, extra))
