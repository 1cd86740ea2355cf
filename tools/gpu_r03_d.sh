#!/bin/bash
# round 3, call D: batched pipeline + stream kernel validation, the w1|w3 / w2 co-residency measurement
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_gpu_batch.py tests/test_pipeline_cabi.py tests/test_gpu_bench.py -m gpu -x -q ) > gpurun_out/r03d_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r03d_tests.log
( time timeout 400 python tools/ffn_overlap.py ) > gpurun_out/r03d_overlap.json 2> gpurun_out/r03d_overlap.err; echo "overlap rc=$?"; cat gpurun_out/r03d_overlap.json; tail -3 gpurun_out/r03d_overlap.err
( time timeout 200 ./tools/mfma_stream_bench part2 ) > gpurun_out/r03d_stream.log 2>&1; echo "stream rc=$?"; tail -32 gpurun_out/r03d_stream.log | cut -c1-220
