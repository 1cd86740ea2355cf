#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
for two in 0 1; do echo "LNB_FAST_GEMM_2WG=$two"; LNB_FAST_GEMM_2WG=$two python tools/prefill_bench.py --modes fast --sizes 512,2048,4096 2>&1 | grep -v amdgpu | cut -c1-140; done
LNB_FAST_GEMM_2WG=1 python -m pytest tests/test_gpu_fast.py -q -m gpu 2>&1 | tail -3
