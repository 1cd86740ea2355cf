#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_fast.py -q -m gpu > gpurun_out/p_fast.log 2>&1; tail -12 gpurun_out/p_fast.log
python bench.py --mode fast --steps 64 --warmup 8 --cpu-steps 0 > gpurun_out/p_bench_fast.json 2>/dev/null; python - <<PY
import json; d=json.load(open("gpurun_out/p_bench_fast.json")); print("fast", d["value"], d["roofline"]["whole_step"]["frac"], {k:v["ms"] for k,v in d["kernels"].items()}, d["config"]["tokens_vs_oracle_golden"])
PY
python bench.py --mode fast --prompt-len 4096 --steps 64 --warmup 8 --cpu-steps 0 > gpurun_out/p_bench_fast_cfg2.json 2>/dev/null; python - <<PY
import json; d=json.load(open("gpurun_out/p_bench_fast_cfg2.json")); print("fast cfg2", d["value"], d["roofline"]["whole_step"]["frac"], {k:v["ms"] for k,v in d["kernels"].items()}, d["prefill"])
PY
python tools/fast_mode_stats.py --seeds 8 --tokens 64 | cut -c1-900
