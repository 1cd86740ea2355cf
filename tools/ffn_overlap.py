#!/usr/bin/env python3
"""What would a w1|w3 -> w2 row-band pipeline have to work with?  (NOTES.md 6.1; run on the GPU box.)
 1. the gate|up kernel cut into half-height row blocks, two per workgroup (LNB_RW_W13=28: rows [0, F/2) complete after the first block of
    every workgroup -- the band order a pipeline needs) against the production 56-row blocks;
 2. the gate|up kernel and the down kernel of a block launched CONCURRENTLY, one workgroup of each on every CU (lnb_profile_kernel 7 / 8):
    the pair against the two launches back to back."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "llama-nuts-and-bolts_amd")]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import lnb
    cfg = dict(lnb.LLAMA_8B, n_layers=6)
    m = lnb.LlamaTransformer(**cfg).fill_synthetic(1234).finalize()
    c = lnb.InferenceContext(m, 320)
    _, t = c.Forward(lnb.synth_tokens(99, 128, cfg["vocab_size"]), 0, want_logits=False)
    c.decode_greedy(t, 128, 8)
    res = {}
    for which, name in ((3, "w1|w3"), (4, "w2"), (7, "pair, w2 launched first"), (8, "pair, w1|w3 launched first"), (6, "whole block")):
        c.profile_kernel(which, 200, 4)
        res[name] = round(1e3 * min(c.profile_kernel(which, 200, 48) for _ in range(3)), 2)
    print(json.dumps(res))
    sys.exit(0)

out = {}
for label, env in (("production (56-row blocks)", {}), ("half-height blocks, two per workgroup (LNB_RW_W13=28)", {"LNB_RW_W13": "28"}),
                   ("56-row blocks, w2 LDS pad 0 (co-residency not forced)", {"LNB_W2_LDS_PAD": "0"})):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    out[label] = json.loads(line[-1]) if line else {"error": r.stderr[-400:]}
print(json.dumps(out, indent=1))
