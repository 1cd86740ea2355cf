#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_pipeline_cabi.py -q -m gpu -x > gpurun_out/o_pipe.log 2>&1; tail -25 gpurun_out/o_pipe.log
