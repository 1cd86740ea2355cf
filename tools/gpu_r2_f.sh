#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_pipeline_cabi.py tests/test_gpu_parity.py -q -m gpu -k "pipe or bench or pipeline" > gpurun_out/f_pipe.log 2>&1; tail -6 gpurun_out/f_pipe.log
python tools/fast_mode_stats.py --seeds 32 --tokens 128 --out gpurun_out/f_fast_stats.json | cut -c1-1500
