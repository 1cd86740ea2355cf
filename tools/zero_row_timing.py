import lnb
cfg = dict(lnb.LLAMA_8B); cfg.update(n_layers=4)
m = lnb.LlamaTransformer(device=0, **cfg).fill_synthetic(7).finalize()
c = lnb.InferenceContext(m, 300)
for which in (0, 3, 5):
    c.profile_kernel(which, 272, 2)
print("all-zero activations:", [round(c.profile_kernel(w, 272, 24) * 1e3, 1) for w in (0, 3, 5)], "us (qkv, w1|w3, head)")
_, tok = c.Forward(lnb.synth_tokens(3, 4, cfg["vocab_size"]), 0, want_logits=False)
c.decode_greedy(tok, 4, 2)
print("real activations:   ", [round(c.profile_kernel(w, 272, 24) * 1e3, 1) for w in (0, 3, 5)], "us")
