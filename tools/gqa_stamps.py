#!/usr/bin/env python3
"""Phase stamps (s_memtime ticks = shader clock cycles, as in the GEMV stamps) of one workgroup of attn_gqa_kernel at 128 sequences of the 8B shape: LNB_ATTN_GQA_DBG=1 python tools/gqa_stamps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llama-nuts-and-bolts_amd"))
import lnb
cfg = dict(lnb.LLAMA_8B); cfg["n_layers"] = 2
n, P = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 160
m = lnb.LlamaTransformer(device=0, **cfg).fill_synthetic(1234).finalize().enable_batch()
ctxs = [lnb.InferenceContext(m, P + 24) for _ in range(n)]
firsts = [c.Forward(lnb.synth_tokens(5 + s, P, cfg["vocab_size"]), 0, want_logits=False)[1] for s, c in enumerate(ctxs)]
b = lnb.Batch(ctxs)
b.decode(firsts, [P] * n, 8)
print("attention us:", round(1e3 * b.profile_kernel(1, P + 8, 32), 2))
lnb.lib().lnbk_attn_gqa_dbg_dump()
