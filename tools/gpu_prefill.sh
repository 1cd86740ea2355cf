#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "long_prefill or matrix_cores" 2>&1 | tail -4
for mf in 1 0; do
LNB_PREFILL_MFMA=$mf timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import lnb, os, time
m = lnb.LlamaTransformer(device=0, **lnb.LLAMA_8B).fill_synthetic(1234).finalize()
c = lnb.InferenceContext(m, 600)
for S in (128, 512) if os.environ["LNB_PREFILL_MFMA"] == "1" else (128,):
    c.reset()
    toks = lnb.synth_tokens(99, S, 128256)
    lnb._chk(lnb.lib().lnb_ctx_synchronize(c.h))
    t0 = time.perf_counter()
    _, tok = c.Forward(toks, 0, want_logits=False)
    dt = time.perf_counter() - t0
    print("LNB_PREFILL_MFMA=%s prefill S=%d: %.1f ms (%.1f TFLOP/s of exact f32 chains), next token %d" % (os.environ["LNB_PREFILL_MFMA"], S, dt * 1e3, 2 * S * 6.98e9 / dt / 1e12, tok), flush=True)
PY
done
