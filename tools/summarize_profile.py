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
    ("gemv_chain_kernel<32, 1,", None, "attn_norm+wqkv+rope GEMV"), ("gemv_quad_kernel<24,", None, "attn_norm+wqkv+rope GEMV"), ("attn_exact_kernel", None, "attention"),
    ("attn_long_scores_kernel", None, "attention (long-context: scores)"), ("attn_long_pv_kernel", None, "attention (long-context: PV)"),

    ("gemv_chain_kernel<56, 2,", None, "ffn_norm+w1|w3+silu GEMV"), ("gemv_chain_kernel<64, 1,", None, "norm+output GEMV"),
    ("fast_gemv_a<1, 1,", None, "attn_norm+wqkv+rope GEMV"), 
    ("fast_gemv_a<2, 3,", None, "ffn_norm+w1|w3+silu GEMV"), ("fast_gemv_a<1, 0,", None, "norm+output GEMV"),
    # batched decode (lnb_batch_kernels.h): mfma_stream_kernel<ACC, EPI>, EPI 0 store / 1 qkv+rope / 2 residual / 3 silu*up
    ("mfma_stream_kernel<1, 1>", None, "batch: wq|wk|wv+rope stream"), ("mfma_stream_kernel<2, 3>", None, "batch: w1|w3+silu stream"),
    ("mfma_stream_kernel<2, 0>", None, "batch: output stream"), ("batch_rmsnorm_xt_kernel", None, "batch: rmsnorm -> B-operand layout"),
]


# wo and w2 run through ONE kernel symbol with the same grid (rocprofv3 reports only static LDS, which is 0 for both): their launches are
# told apart by size -- durations in the trace pass, bytes in the PMC pass -- at the geometric mean of the extremes (K = 4096 against 14336)
SPLIT = {"rowcast_kernel<2>": ("wo+residual GEMV", "w2+residual GEMV"), "rowcast_lds_kernel<2>": ("wo+residual GEMV", "w2+residual GEMV"), "fast_gemv_b<2>": ("wo+residual GEMV", "w2+residual GEMV"),
         "mfma_stream_kernel<1, 2>": ("batch: wo+residual stream", "batch: w2+residual stream")}


def split_two(vals):
    lo, hi = min(vals), max(vals)
    thr = (lo * hi) ** 0.5
    return [v for v in vals if v <= thr], [v for v in vals if v > thr]


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
    cmdline = ("python tools/batch_bench.py --n 16 --steps 16` (batched exact decode, 16 sequences; prefill of the 16 prompts included" if "batch" in label else
               "python bench.py%s%s --steps 32 --warmup 4`" % (" --mode fast" if "fast" in label else "", " --prompt-len 4096 (configs[2] decode: long-context attention kernels)" if "cfg2" in label else ""))
    lines = ["# %s: rocprofv3 summary of `%s (Llama-3.1-8B shape, 1 x MI355X)" % (label, cmdline), "",
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
        if cls and len(d) > 30:                   # decode-sized launches (one per layer per token), not the prefill's
            c = classes.setdefault(cls, {"kernel": kn[:96], "launches": 0, "avg_us_under_rocprof": 0.0})
            if len(d) > c["launches"]:
                c.update(kernel=kn[:96], launches=len(d), avg_us_under_rocprof=round(avg, 2), grid_threads=int(grid), lds_bytes=lds)
                if mb:
                    c["hbm_read_bytes_per_launch"] = int(mb * 1024 * 1024)
        lines.append("| %s | %s | %s | %s | %d | %.1f | %s | %s |" % (kn.replace("|", "/")[:110], cls or "", grid, lds, len(d), avg, ("%.1f" % mb) if mb else "-",
                                                                     ("%.0f" % (mb * 1.048576 / avg * 1e3)) if mb else "-"))
    for pat, (c_lo, c_hi) in SPLIT.items():
        for key in per:
            if pat in key[0] and len(per[key]) > 200:
                d_lo, d_hi = split_two(per[key])
                f_lo, f_hi = split_two(fetch[key]) if fetch.get(key) else ([], [])
                for cls, dd, ff in ((c_lo, d_lo, f_lo), (c_hi, d_hi, f_hi)):
                    if dd:
                        classes[cls] = {"kernel": key[0][:96], "launches": len(dd), "avg_us_under_rocprof": round(sum(dd) / len(dd), 2), "grid_threads": int(key[1]),
                                        "split": "by launch size (see tools/summarize_profile.py)"}
                        if ff:
                            classes[cls]["hbm_read_bytes_per_launch"] = int(2 * sum(ff) / len(ff) * 1024)
                        lines.append("| %s | %s | %s | - | %d | %.1f | %s | %s |" % (key[0][:60], cls.replace("|", "/"), key[1], len(dd), sum(dd) / len(dd),
                                     ("%.1f" % (2 * sum(ff) / len(ff) / 1024.0)) if ff else "-", ("%.0f" % (2 * sum(ff) / len(ff) * 1024 / (sum(dd) / len(dd)) / 1e3)) if ff else "-"))
    for fn in ("trace_bench.json", "bench_default.json", "batch_bench.json"):
        pth = os.path.join(src, fn)
        if os.path.exists(pth) and os.path.getsize(pth):
            lines += ["", "## %s" % fn, "```json", open(pth).read().strip(), "```"]
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE pass of bench.py (tools/gpu_profile_r02.sh), KB x 2 x 1024: gfx950 correction of MI355X_MICROARCH.md",
               "git_head": git_head(), "classes": {k: v for k, v in classes.items() if "hbm_read_bytes_per_launch" in v}},
              open(os.path.join(DST, label + "_traffic.json"), "w"), indent=1)
    extra = os.path.join(DST, label + "_before_after.md")      # hand-kept before/after table of the round: appended to the generated summary
    tail = open(extra).read() if os.path.exists(extra) else ""
    open(os.path.join(DST, label + "_summary.md"), "w").write("\n".join(lines) + "\n" + tail)
    print("\n".join(lines[:30]))


for lab in sys.argv[1:] or ["r02"]:
    one(lab)
