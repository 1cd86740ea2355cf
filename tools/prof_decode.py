#!/usr/bin/env python3
"""Small driver for rocprofv3: builds the synthetic 8B model, prefills a few tokens and launches every kernel
class of the decode step a few times (eager, cycling through the layers) plus a short graph-replayed decode."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llama-nuts-and-bolts_amd"))
import lnb  # noqa: E402

iters = int(os.environ.get("PROF_ITERS", "8"))
steps = int(os.environ.get("PROF_STEPS", "4"))
pos = int(os.environ.get("PROF_POS", "160"))
m = lnb.LlamaTransformer(device=0, **lnb.LLAMA_8B).fill_synthetic(1234).finalize()
c = lnb.InferenceContext(m, 512)
_, tok = c.Forward(lnb.synth_tokens(99, 16, 128256), 0, want_logits=False)
for which in range(6):
    ms = c.profile_kernel(which, pos, iters)
    print("kernel %d: %.4f ms" % (which, ms), flush=True)
if steps:
    out, ms = c.decode_greedy(tok, 16, steps)
    print("decode %d steps: %.3f ms/step" % (steps, ms / steps))
c.close(); m.close()
