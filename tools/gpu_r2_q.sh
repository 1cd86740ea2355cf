#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( time python -m pytest tests/ -q -m gpu -x ) > gpurun_out/q_gpu_suite.log 2>&1; tail -5 gpurun_out/q_gpu_suite.log
python bench.py --prompt-len 4096 --steps 64 --warmup 8 --cpu-steps 0 > gpurun_out/q_bench_cfg2.json 2>/dev/null; python - <<PY
import json; d=json.load(open("gpurun_out/q_bench_cfg2.json")); print("cfg2", d["value"], d["roofline"]["whole_step"]["frac"], {k:v["ms"] for k,v in d["kernels"].items()})
PY
python tools/att_timing.py 2>&1 | grep -v amdgpu
