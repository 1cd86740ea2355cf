#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_fast.py -q -m gpu -x 2>&1 | tail -3
python tools/prefill_bench.py --out gpurun_out/prof_r02/prefill.json 2>&1 | tail -9
