#!/bin/bash
# round 2, GPU call C: long-context decode attention -- parity, crossover timing, configs[2] decode bench
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_configs.py -q -m gpu -x > gpurun_out/c_configs.log 2>&1; tail -12 gpurun_out/c_configs.log
python tools/att_timing.py 2>&1 | grep -v amdgpu.ids
python bench.py --prompt-len 4096 --steps 64 --warmup 8 --cpu-steps 0 > gpurun_out/c_bench_cfg2.json 2> gpurun_out/c_bench_cfg2.err; python - <<PY
import json; d=json.load(open("gpurun_out/c_bench_cfg2.json")); print("cfg2", d["value"], d["roofline"]["whole_step"], {k:v["ms"] for k,v in d["kernels"].items()}, d["prefill"])
PY
tail -3 gpurun_out/c_bench_cfg2.err
