#!/bin/bash
# prefill check: GEMM parity tests + the 8B 512-token MFMA == row-by-row digest, then prefill timings on the 8B shape
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_8b.py -x -q -m gpu -k "prefill or matrix_cores or mfma or linear or checkpoint" 2>&1 | tail -4
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import lnb, os, time
m = lnb.LlamaTransformer(device=0, **lnb.LLAMA_8B).fill_synthetic(1234).finalize()
c = lnb.InferenceContext(m, 2100)
for S in (128, 512, 2048):
    for rep in range(2):
        c.reset()
        toks = lnb.synth_tokens(99, S, 128256)
        lnb._chk(lnb.lib().lnb_ctx_synchronize(c.h))
        t0 = time.perf_counter()
        _, tok = c.Forward(toks, 0, want_logits=False)
        dt = time.perf_counter() - t0
    print("prefill S=%d: %.1f ms (%.1f TFLOP/s of exact f32 chains), next token %d" % (S, dt * 1e3, 2 * S * 6.98e9 / dt / 1e12, tok), flush=True)
PY
