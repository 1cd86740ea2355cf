#!/usr/bin/env python3
"""Condense gpurun_out/prof_<label>/ (rocprofv3 --kernel-trace --stats and --pmc passes of bench.py, tools/gpu_profile_r02.sh) into
profiles/<label>_*.   Usage: python tools/summarize_profile.py r02 [r02_fast ...]

Writes  profiles/<label>_rocprofv3_kernel_stats.csv   rocprofv3's own --stats table of the bench command
        profiles/<label>_summary.md                   per kernel (name, grid, LDS): launches, average duration, corrected HBM read bytes
        profiles/<label>_traffic.json                 the same per decode kernel CLASS (the names bench.py prints), + the git head
HBM read bytes = FETCH_SIZE (KB) x 2 x 1024: the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md, section HBM (wide
coalesced reads are tallied at half their bytes); collected in its own --pmc pass."""
import collections
import csv
import json
import os
import shutil
import subprocess
import sys

DST = "profiles"
os.makedirs(DST, exist_ok=True)

# decode kernel classes of the 8B shape: (substring of the kernel name, LDS bytes or None) -> class name of bench.py
CLASSES = [
    ("gemv_chain_kernel<32, 1,", None, "attn_norm+wqkv+rope GEMV"), ("attn_exact_kernel", None, "attention"),
    ("attn_long_scores_kernel", None, "attention (long-context: scores)"), ("attn_long_pv_kernel", None, "attention (long-context: PV)"),
    ("rowcast_kernel<2>", 16384, "wo+residual GEMV"), ("rowcast_kernel<2>", 57344, "w2+residual GEMV"),
    ("gemv_chain_kernel<56, 2,", None, "ffn_norm+w1|w3+silu GEMV"), ("gemv_chain_kernel<64, 1,", None, "norm+output GEMV"),
    ("fast_gemv_a<1, 1,", None, "attn_norm+wqkv+rope GEMV"), ("fast_gemv_b<2>", 8256, "wo+residual GEMV"), ("fast_gemv_b<2>", 28736, "w2+residual GEMV"),
    ("fast_gemv_a<2, 3,", None, "ffn_norm+w1|w3+silu GEMV"), ("fast_gemv_a<1, 0,", None, "norm+output GEMV"),
]


def classify(name, lds):
    for pat, l, cls in CLASSES:
        if pat in name and (l is None or str(l) == str(lds)):
            return cls
    return None


def git_head():
    try:
        return subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        return None


def one(label):
    src = os.path.join("gpurun_out", "prof_" + label)
    stats = os.path.join(src, "trace", "trace_kernel_stats.csv")
    if os.path.exists(stats):
        shutil.copy(stats, os.path.join(DST, label + "_rocprofv3_kernel_stats.csv"))
    trace = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_trace.csv"))))
    per = collections.defaultdict(list)
    for r in trace:
        grid = str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
        per[(r["Kernel_Name"], grid, r.get("LDS_Block_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
    fetch = collections.defaultdict(list)
    fp = os.path.join(src, "pmc_fetch", "pmc_counter_collection.csv")
    if os.path.exists(fp):
        for r in csv.DictReader(open(fp)):
            if r["Counter_Name"] == "FETCH_SIZE":
                fetch[(r["Kernel_Name"], r["Grid_Size"], r.get("LDS_Block_Size", ""))].append(float(r["Counter_Value"]))
    lines = ["# %s: rocprofv3 summary of `python bench.py%s --steps 32 --warmup 4` (Llama-3.1-8B shape, 1 x MI355X)" % (label, " --mode fast" if "fast" in label else ""), "",
             "Per-dispatch averages by (kernel, grid, LDS).  `HBM read` = FETCH_SIZE (KB) x 2 / 1024 -- the gfx950 correction of",
             "/opt/skills/guides/MI355X_MICROARCH.md section HBM (wide coalesced reads are tallied at half their bytes); its own --pmc pass.", "",
             "| kernel | class | grid (threads) | LDS B | launches | avg us | HBM read MB (PMC, corrected) | GB/s |", "|---|---|---|---|---|---|---|---|"]
    classes = {}
    for key in sorted(per, key=lambda k: -sum(per[k])):
        kn, grid, lds = key
        d = per[key]
        if len(d) < 8 and sum(d) < 2000:
            continue
        avg = sum(d) / len(d)
        f = fetch.get(key)
        mb = (2 * sum(f) / len(f) / 1024.0) if f else None
        cls = classify(kn, lds)
        if cls and len(d) > 100:                   # decode-sized launches (one per layer per token), not the prefill's
            c = classes.setdefault(cls, {"kernel": kn[:96], "launches": 0, "avg_us_under_rocprof": 0.0})
            if len(d) > c["launches"]:
                c.update(kernel=kn[:96], launches=len(d), avg_us_under_rocprof=round(avg, 2), grid_threads=int(grid), lds_bytes=lds)
                if mb:
                    c["hbm_read_bytes_per_launch"] = int(mb * 1024 * 1024)
        lines.append("| %s | %s | %s | %s | %d | %.1f | %s | %s |" % (kn.replace("|", "/")[:110], cls or "", grid, lds, len(d), avg, ("%.1f" % mb) if mb else "-",
                                                                     ("%.0f" % (mb * 1.048576 / avg * 1e3)) if mb else "-"))
    for fn in ("trace_bench.json", "bench_default.json"):
        pth = os.path.join(src, fn)
        if os.path.exists(pth) and os.path.getsize(pth):
            lines += ["", "## %s" % fn, "```json", open(pth).read().strip(), "```"]
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE pass of bench.py (tools/gpu_profile_r02.sh), KB x 2 x 1024: gfx950 correction of MI355X_MICROARCH.md",
               "git_head": git_head(), "classes": {k: v for k, v in classes.items() if "hbm_read_bytes_per_launch" in v}},
              open(os.path.join(DST, label + "_traffic.json"), "w"), indent=1)
    open(os.path.join(DST, label + "_summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:30]))


for lab in sys.argv[1:] or ["r02"]:
    one(lab)
