#!/bin/bash
# round 3, call N: upper bound of fusing the token kernels (argmax / embedding gather) into their neighbours: the step without them
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd TMPDIR=/tmp
cat > /tmp/tk.py <<'PY'
import lnb, sys, time
m = lnb.LlamaTransformer(device=0, **lnb.LLAMA_8B).fill_synthetic(1234).finalize()
c = lnb.InferenceContext(m, 600)
_, tok = c.Forward(lnb.synth_tokens(99, 128, 128256), 0, want_logits=False)
c.decode_greedy(tok, 128, 8)
best = 1e9
for rep in range(3):
    out, ms = c.decode_greedy(tok, 136, 256)
    best = min(best, ms / 256)
print("LNB_MEASURE_SKIP_TOKEN_KERNELS=%s: %.4f ms per step = %.2f tokens/s (HIP events, 256 steps, context 136..392)" % (sys.argv[1], best, 1e3 / best))
PY
for rep in 1 2; do for sk in 0 1; do LNB_MEASURE_SKIP_TOKEN_KERNELS=$sk timeout 300 python /tmp/tk.py $sk 2>&1 | tail -2; done; done | tee gpurun_out/r03n_token_kernels.log
