cd "$GRAFT_REPO_ROOT"; export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
for t in 0 64; do
LNB_GEMM_TILE=$t timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import lnb, os, time
m = lnb.LlamaTransformer(device=0, **lnb.LLAMA_8B).fill_synthetic(1234).finalize()
c = lnb.InferenceContext(m, 2100)
for S in (1024, 2048):
    for rep in range(2):
        c.reset(); toks = lnb.synth_tokens(99, S, 128256)
        lnb._chk(lnb.lib().lnb_ctx_synchronize(c.h)); t0 = time.perf_counter()
        _, tok = c.Forward(toks, 0, want_logits=False); dt = time.perf_counter() - t0
    print("LNB_GEMM_TILE=%s prefill S=%d: %.1f ms, next token %d" % (os.environ["LNB_GEMM_TILE"], S, dt * 1e3, tok), flush=True)
PY
done
