#!/bin/bash
# first GPU session: microbench, sanity, parity tests, short bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
echo "== host: $(nproc) cores, $(free -g | awk '/Mem/{print $2}') GB" | tee gpurun_out/host.txt
rocm-smi --showproductname 2>/dev/null | head -8 >> gpurun_out/host.txt
timeout 120 tools/microbench > gpurun_out/microbench.txt 2>&1; echo "microbench rc=$?"
timeout 180 python - > gpurun_out/sanity.txt 2>&1 <<'PY'
import numpy as np, lnb
from oracle import oracle as orc
rng = np.random.default_rng(0)
for (rows,n,k,rw) in [(1,64,64,16),(1,256,256,16),(1,256,256,64),(2,100,896,32)]:
    x = orc.f32_to_bf16(rng.standard_normal((rows,k)).astype(np.float32)); w = orc.f32_to_bf16((rng.standard_normal((n,k))*0.05).astype(np.float32))
    y = lnb.op_linear(x, w, rw=rw)
    ref = np.zeros_like(y); orc.lib().orc_linear_bf16(orc._p(x), orc._p(w), orc._p(ref), rows, n, k, 4)
    print(rows,n,k,rw,"mismatches:", int((y!=ref).sum()), "of", y.size, flush=True)
PY
echo "sanity rc=$?"; tail -5 gpurun_out/sanity.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/pytest_parity.txt 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_parity.txt
timeout 600 python bench.py --steps 64 --warmup 8 --cpu-steps 0 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; echo "bench rc=$?"; tail -3 gpurun_out/bench1.err; cat gpurun_out/bench1.json
