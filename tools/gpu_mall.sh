#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
for same in 0 1; do
LNB_PROFILE_SAME_LAYER=$same timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu.ids
import lnb, os
m = lnb.LlamaTransformer(device=0, **lnb.LLAMA_8B).fill_synthetic(1234).finalize()
c = lnb.InferenceContext(m, 512)
_, tok = c.Forward(lnb.synth_tokens(99, 16, 128256), 0, want_logits=False)
print("same layer =", os.environ["LNB_PROFILE_SAME_LAYER"])
for which in (0, 2, 3, 4):
    print("  kernel %d: %.2f us" % (which, 1000 * c.profile_kernel(which, 200, 32)), flush=True)
PY
done
