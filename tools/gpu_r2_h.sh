#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_fast.py -q -m gpu > gpurun_out/h_fast.log 2>&1; tail -25 gpurun_out/h_fast.log
python tools/prefill_bench.py --modes fast --out gpurun_out/h_prefill.json 2>&1 | tail -5
