#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_configs.py -q -m gpu -x -k "long_context or crossover" > gpurun_out/e_configs.log 2>&1; tail -4 gpurun_out/e_configs.log
python tools/att_timing.py 2>&1 | grep -v amdgpu.ids
