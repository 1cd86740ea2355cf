#!/bin/bash
# round 2, GPU call B: tolerance-mode tests incl. the bf16 GEMM, prefill timings in both modes, fast row-group check, exact norm-kernel wave timing
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_fast.py -q -m gpu > gpurun_out/b_fast.log 2>&1; tail -30 gpurun_out/b_fast.log
python tools/prefill_bench.py --out gpurun_out/b_prefill.json 2>&1 | tail -12
python bench.py --mode fast --steps 64 --warmup 8 --cpu-steps 0 > gpurun_out/b_bench_fast.json 2>/dev/null; python - <<PY
import json; d=json.load(open("gpurun_out/b_bench_fast.json")); print("auto", d["value"], d["roofline"]["whole_step"]["frac"], {k:v["ms"] for k,v in d["kernels"].items()}, d["prefill"])
PY
LNB_FAST_RG=64 python bench.py --mode fast --steps 32 --warmup 4 --cpu-steps 0 > gpurun_out/b_bench_fast_rg64.json 2>/dev/null; python - <<PY
import json; d=json.load(open("gpurun_out/b_bench_fast_rg64.json")); print("RG 64", d["value"], {k:v["ms"] for k,v in d["kernels"].items()})
PY
LNB_GEMV_TIMING=1 PROF_ITERS=8 PROF_STEPS=0 python tools/prof_decode.py 2>&1 | grep -v amdgpu.ids | tail -60
