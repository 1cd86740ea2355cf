#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<round>/ directory (rocprofv3 --kernel-trace --stats and --pmc passes of bench.py) into
profiles/<round>_*.  Usage: python tools/summarize_profile.py r01"""
import collections
import csv
import json
import os
import shutil
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
SRC = os.path.join("gpurun_out", "prof_" + R)
DST = "profiles"
os.makedirs(DST, exist_ok=True)
shutil.copy(os.path.join(SRC, "trace", "trace_kernel_stats.csv"), os.path.join(DST, R + "_rocprofv3_kernel_stats.csv"))

NAMES = {"gemv_chain_kernel<32, 1, 12288, 6, 5, 1, true>": "attn_norm+wq|wk|wv+RoPE+KV (thin, RW=32, exact parallel norm sum)",
         "rowcast_kernel<2>": "wo / w2 + residual (row-broadcast DPP chain, 4 rows per wave)",
         "rowcast_kernel<0>": "plain linear (row-broadcast DPP chain)",
         "gemv_chain_kernel<56, 2, 14336, 7, 8, 3, true>": "ffn_norm+w1|w3+SiLU*up (fat, 256 blocks of 56 rows x 2 chains, v_pk_add_f32)",
         "gemm_mfma_kernel": "prefill GEMM on the f32 matrix cores (exact order)", "rmsnorm_rows_kernel": "prefill RMSNorm (one wave per row)",
         "gemv_chain_kernel<64, 1, 12288, 6, 8, 0, true>": "norm+output (fat, RW=64)",
         "attn_exact_kernel<128>": "attention (scores, f64 softmax, PV)",
         "attn_mfma_kernel<128>": "prefill attention on the f32 matrix cores (16 query rows per wave, exact order)"}
trace = list(csv.DictReader(open(os.path.join(SRC, "trace", "trace_kernel_trace.csv"))))
per = collections.defaultdict(list)
for r in trace:
    grid = str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
    per[(r["Kernel_Name"], grid, r.get("LDS_Block_Size", ""))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
fetch = collections.defaultdict(list)
for r in csv.DictReader(open(os.path.join(SRC, "pmc_fetch", "pmc_counter_collection.csv"))):
    if r["Counter_Name"] == "FETCH_SIZE":
        fetch[(r["Kernel_Name"], r["Grid_Size"], r.get("LDS_Block_Size", ""))].append(float(r["Counter_Value"]))
sq = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(os.path.join(SRC, "pmc_sq", "pmc_counter_collection.csv"))):
    sq[(r["Kernel_Name"], r["Grid_Size"], r.get("LDS_Block_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = ["# %s: rocprofv3 summary of `python bench.py --steps 32 --warmup 4` (Llama-3.1-8B shape, 1 x MI355X)" % R, "",
         "Per-dispatch averages of the DECODE-sized launches (one token; the prefill launches of the same kernels are listed",
         "separately by their larger grid).  `HBM read` = FETCH_SIZE (KB) x 2 / 1024 -- the gfx950 correction of",
         "/opt/skills/guides/MI355X_MICROARCH.md section HBM (wide coalesced reads are tallied at half their bytes).", "",
         "| kernel | grid (threads) | LDS B | launches | avg us | HBM read MB (PMC, corrected) | GB/s | WAVE_CYCLES busy/wait (quad-cycles per launch) |", "|---|---|---|---|---|---|---|---|"]
traffic = {}
for key in sorted(per, key=lambda k: -sum(per[k])):
    kn, grid, lds = key
    if "gemv" not in kn and "attn" not in kn and "argmax" not in kn and "rowcast" not in kn and "gemm_mfma" not in kn and "rmsnorm_rows" not in kn:
        continue
    label = kn
    for pat, nm in NAMES.items():
        if pat in kn:
            label = "%s — %s" % (pat, nm)
    d = per[key]
    avg = sum(d) / len(d)
    f = fetch.get(key)
    mb = (2 * sum(f) / len(f) / 1024.0) if f else None
    if mb and len(d) > 200:                       # decode-sized launches only (one per layer per token)
        traffic[label.split(" — ")[0]] = {"hbm_read_bytes_per_launch": int(mb * 1024 * 1024), "avg_us_under_rocprof": round(avg, 2), "launches": len(d)}
    s = sq.get(key)
    sqtxt = ""
    if s:
        g = lambda n: sum(s[n]) / len(s[n]) if s[n] else 0
        sqtxt = "active %.2e / wait_any %.2e / waves %d" % (g("SQ_ACTIVE_INST_ANY"), g("SQ_WAIT_ANY"), g("SQ_WAVES"))
    lines.append("| %s | %s | %s | %d | %.1f | %s | %s | %s |" % (label.replace("|", "/"), grid, lds, len(d), avg, ("%.1f" % mb) if mb else "-",
                                                                   ("%.0f" % (mb / avg * 1e3)) if mb else "-", sqtxt))
for fn in ("trace_bench.json", "bench_default.json"):
    p = os.path.join(SRC, fn)
    if os.path.exists(p) and os.path.getsize(p):
        lines += ["", "## %s" % fn, "```json", open(p).read().strip(), "```"]
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE pass of bench.py (tools/gpu_profile.sh), KB x 2 x 1024: gfx950 correction of MI355X_MICROARCH.md",
           "kernels": traffic}, open(os.path.join(DST, R + "_traffic.json"), "w"), indent=1)
open(os.path.join(DST, R + "_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:40]))
