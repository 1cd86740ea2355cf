#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( time python -m pytest tests/ -q -m gpu -x ) > gpurun_out/k_gpu_suite.log 2>&1; tail -6 gpurun_out/k_gpu_suite.log
