#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
python -m pytest tests/test_gpu_fast.py -q -m gpu > gpurun_out/i_fast.log 2>&1; tail -5 gpurun_out/i_fast.log
python tools/prefill_bench.py --modes fast --out gpurun_out/i_prefill.json 2>&1 | tail -5
python bench.py --mode fast --steps 64 --warmup 8 --cpu-steps 0 > gpurun_out/i_bench_fast.json 2>/dev/null; python - <<PY
import json; d=json.load(open("gpurun_out/i_bench_fast.json")); print("fast", d["value"], d["roofline"]["whole_step"]["frac"], {k:v["ms"] for k,v in d["kernels"].items()}, d["prefill"]["ms"])
PY
cat > /tmp/at.py <<'PY'
import lnb
cfg = dict(lnb.LLAMA_8B); cfg.update(n_layers=2)
m = lnb.LlamaTransformer(device=0, **cfg).fill_synthetic(1234).finalize(rope_rows=8192)
c = lnb.InferenceContext(m, 4400).set_attention(0, 0)
print(c.profile_kernel(1, 4100, 32))
PY
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/attprof -o t -- python /tmp/at.py > /dev/null 2>&1; grep -i "attn_long" /tmp/attprof/*/t_kernel_stats.csv /tmp/attprof/t_kernel_stats.csv 2>/dev/null | cut -c1-300
