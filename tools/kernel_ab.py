#!/usr/bin/env python3
"""A/B aid: per-kernel decode timings (lnb_profile_kernel) of the 8B block shape on a 4-layer model -- seconds per library variant.
usage: LNB_SO=/path/to/variant.so python tools/kernel_ab.py [iters]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llama-nuts-and-bolts_amd"))
import lnb

cfg = dict(lnb.LLAMA_8B); cfg["n_layers"] = int(os.environ.get("AB_LAYERS", "4"))     # (AB_LAYERS=32: the launches cycle through the full model's weights, as a decode step does)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
m = lnb.LlamaTransformer(device=0, **cfg).fill_synthetic(1234)
if os.environ.get("AB_ZERO"):     # DVFS experiment: the blocks' matrices zeroed (AB_ZERO=1) -- same instruction stream, less switching; do the chain-bound launches clock higher?
    import numpy as np
    for name, shape in m.tensor_infos():
        if name.startswith("layers.") and ("attention.w" in name or "feed_forward.w" in name):
            m.set_tensor(name, np.zeros(shape, dtype=np.uint16))
m.finalize(rope_rows=max(0, int(os.environ.get("AB_POS", "272")) + 64))
POS = int(os.environ.get("AB_POS", "272"))          # context length the attention is timed at (AB_POS=4100: configs[2]'s decode)
c = lnb.InferenceContext(m, max(512, POS + 64))
prompt = lnb.synth_tokens(99, 128, cfg["vocab_size"])
_, first = c.Forward(prompt, 0, want_logits=False)
toks, _ = c.decode_greedy(first, 128, 16)
names = ["qkv", "attn", "wo", "w13", "w2", "head", "block"]
if os.environ.get("AB_SCHED"):
    c.set_schedule(os.environ["AB_SCHED"])
out, reps = {}, {}
for rep in range(3):
    for w, n in enumerate(names):
        ms = c.profile_kernel(w, POS, iters)
        out[n] = min(out.get(n, 1e9), round(ms * 1e3, 2))
        reps.setdefault(n, []).append(round(ms * 1e3, 2))
print(json.dumps({"so": os.path.basename(os.environ.get("LNB_SO", "default")), "layers": cfg["n_layers"], "pos": POS, "us": out, "reps": reps, "tok": [int(t) for t in toks[-3:]]}))
