"""decode attention time per layer at several contexts, both kernel forms (8B head geometry, two layers)
    python tools/att_timing.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llama-nuts-and-bolts_amd"))
import lnb
cfg = dict(lnb.LLAMA_8B); cfg.update(n_layers=2)
m = lnb.LlamaTransformer(device=0, **cfg).fill_synthetic(1234).finalize(rope_rows=8192)
c = lnb.InferenceContext(m, 4400)
for pos in (63, 127, 271, 383, 511, 767, 1023, 2047, 4100):
    row = []
    for thr, z, lazy in ((10 ** 9, 0, "1"), (0, 2, "0"), (0, 2, "1"), (0, 8, "1"), (0, 3, "1")):
        if thr > 0 and pos + 1 > 7000:
            row.append(float("nan")); continue
        os.environ["LNB_ATTN_LAZY"] = lazy
        c.set_attention(thr, z)
        row.append(c.profile_kernel(1, pos, 32) * 1e3)
    print("attention at T=%5d: one workgroup per head %7.2f us | two launches, r02 PV kernel %7.2f us | two launches, lazily certified PV %7.2f us | ONE launch %7.2f us | two launches, serial Z walk %7.2f us" % (pos + 1, *row), flush=True)
