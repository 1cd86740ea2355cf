#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_pipeline_cabi.py -q -m gpu -x -k "one_stage" 2>&1 | tail -15
