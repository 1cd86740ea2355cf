#!/bin/bash
# round 2, GPU call D: the whole -m gpu suite, attention timing, configs[2] decode bench, one-GPU pipeline tick path (host time per tick)
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( time python -m pytest tests/ -q -m gpu -x ) > gpurun_out/d_gpu_suite.log 2>&1; tail -15 gpurun_out/d_gpu_suite.log
python tools/att_timing.py 2>&1 | grep -v amdgpu.ids
python bench.py --prompt-len 4096 --steps 64 --warmup 8 --cpu-steps 0 > gpurun_out/d_bench_cfg2.json 2> gpurun_out/d_bench_cfg2.err; python - <<PY
import json; d=json.load(open("gpurun_out/d_bench_cfg2.json")); print("cfg2", d["value"], d["roofline"]["whole_step"]["frac"], {k:v["ms"] for k,v in d["kernels"].items()})
PY
LNB_FORCE_PIPELINE=1 MASTER_PORT=29578 python bench.py --gpus 1 --steps 32 --warmup 4 > gpurun_out/d_bench_pipe1.json 2> gpurun_out/d_bench_pipe1.err; cat gpurun_out/d_bench_pipe1.json; tail -3 gpurun_out/d_bench_pipe1.err
LNB_FORCE_PIPELINE=1 LNB_PIPELINE_GRAPH=0 MASTER_PORT=29579 python bench.py --gpus 1 --steps 32 --warmup 4 > gpurun_out/d_bench_pipe1_nograph.json 2>/dev/null; cat gpurun_out/d_bench_pipe1_nograph.json
