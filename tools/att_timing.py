import lnb, os
cfg = dict(lnb.LLAMA_8B); cfg.update(n_layers=2)
m = lnb.LlamaTransformer(device=0, **cfg).fill_synthetic(1234).finalize(rope_rows=8192)
c = lnb.InferenceContext(m, 4400)
for pos in (271, 1023, 2047, 4100):
    ms = c.profile_kernel(1, pos, 32)
    print("attention at T=%d: %.2f us" % (pos + 1, ms * 1e3), flush=True)
