#!/bin/bash
# round 3, call A: issue-rate microbenchmarks + the streaming-kernel prototype, then the new parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
mkdir -p gpurun_out
( time timeout 600 ./tools/mfma_stream_bench ) > gpurun_out/r03a_stream.log 2>&1; echo "stream rc=$?"
( time timeout 1500 python -m pytest tests/test_gpu_round3.py -x -q ) > gpurun_out/r03a_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r03a_tests.log
tail -70 gpurun_out/r03a_stream.log
