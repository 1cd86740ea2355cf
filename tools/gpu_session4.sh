#!/bin/bash
# parity + short bench + attention/argmax timing vs context length
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/pytest_parity.txt 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_parity.txt
timeout 600 python bench.py --steps 64 --warmup 8 --cpu-steps 0 > gpurun_out/bench2.json 2> gpurun_out/bench2.err; echo "bench rc=$?"; tail -3 gpurun_out/bench2.err; python -c "
import json; d=json.load(open('gpurun_out/bench2.json')); print(d['value'], d['ms_per_step']); [print(k, v) for k,v in d['kernels'].items()]"
timeout 600 python - <<'PY'
import lnb
m = lnb.LlamaTransformer(device=0, **lnb.LLAMA_8B).fill_synthetic(1234).finalize()
c = lnb.InferenceContext(m, 2048)
_, tok = c.Forward(lnb.synth_tokens(99, 16, 128256), 0, want_logits=False)
for pos in (15, 63, 127, 255, 511, 1023, 2040):
    print("attention T=%d: %.2f us" % (pos + 1, 1000 * c.profile_kernel(1, pos, 32)))
out, ms = c.decode_greedy(tok, 16, 32)
out, ms = c.decode_greedy(int(out[-1]), 48, 64)
print("decode @T~80: %.3f ms/step" % (ms / 64))
PY
