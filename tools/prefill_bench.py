#!/usr/bin/env python3
"""Prefill (one Forward over S prompt rows, last-row logits only) of the 8B shape in both arithmetic modes.
exact: the reference's k-ordered chains on the f32 matrix cores (peak 157.3 TFLOP/s); fast: bf16 matrix cores (peak 2500 TFLOP/s).
FLOPs counted = 2 x S x (layer matmul elements) + the LM head row; attention FLOPs are not counted (so the rates are lower bounds).
    python tools/prefill_bench.py [--sizes 128,512,2048,4096] [--modes exact,fast] [--out gpurun_out/prefill.json]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llama-nuts-and-bolts_amd"))
import lnb  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="128,512,2048,4096")
ap.add_argument("--modes", default="exact,fast")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--out", default="")
ap.add_argument("--layers", type=int, default=32, help="blocks of the model (fewer: a cheap run under the profiler -- the launches per block are the same)")
ap.add_argument("--stream", action="store_true", help="enable the matrix-core copy of the weights first: exact prefill products run on gemm_stream_kernel")
a = ap.parse_args()
sizes = [int(s) for s in a.sizes.split(",")]
m = lnb.LlamaTransformer(device=0, **dict(lnb.LLAMA_8B, n_layers=a.layers)).fill_synthetic(1234).finalize(rope_rows=max(sizes) + 64)
if a.stream:
    m.enable_batch()
c = lnb.InferenceContext(m, max(sizes) + 8)
MATMUL = 6979321856 // 32 * a.layers      # weight elements of the blocks (multiply-accumulates per row)
res = []
for mode in a.modes.split(","):
    c.set_mode(mode)
    for S in sizes:
        toks = lnb.synth_tokens(99, S, 128256)
        best = 1e9
        for rep in range(a.reps):
            c.reset()
            lnb._chk(lnb.lib().lnb_ctx_synchronize(c.h))
            t0 = time.perf_counter()
            _, tok = c.Forward(toks, 0, want_logits=False)
            best = min(best, time.perf_counter() - t0)
        tf = 2.0 * S * MATMUL / best / 1e12
        peak = 157.3 if mode == "exact" else 2500.0
        r = {"mode": mode, "kernel": ("gemm_stream_kernel (weights M16 -> A operand)" if a.stream else "gemm_mfma_kernel (LDS-tiled)" if os.environ.get("LNB_PREFILL_NATIVE") == "0" else
                        "gemm_stream_kernel (resident layouts -> A operand, no second copy)") if mode == "exact" else "fast_gemm_kernel", "rows": S, "ms": round(best * 1e3, 2), "TFLOP/s": round(tf, 1), "peak_TFLOP/s": peak, "frac_of_peak": round(tf / peak, 4),
             "frac_of_bf16_peak_2500": round(tf / 2500.0, 4), "next_token": int(tok)}
        print(json.dumps(r), flush=True)
        res.append(r)
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
