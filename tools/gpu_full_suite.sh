#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
nproc > gpurun_out/k_gpu_suite.log
( time python -m pytest tests/ -q -m gpu -x --durations=25 ) >> gpurun_out/k_gpu_suite.log 2>&1; tail -45 gpurun_out/k_gpu_suite.log
