#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
LNB_GEMV_TIMING=1 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import lnb
cfg = dict(lnb.LLAMA_8B); cfg.update(n_layers=2)
m = lnb.LlamaTransformer(device=0, **cfg).fill_synthetic(1234).finalize(rope_rows=8192)
c = lnb.InferenceContext(m, 4400).set_attention(0, 0)
print(c.profile_kernel(1, 4100, 16)); print(c.profile_kernel(1, 2047, 16))
PY
