#!/bin/bash
# round 3, call B: batched decode parity + bench line with the batched section, prototype ablations, then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_batch.py -x -q -s ) > gpurun_out/r03b_batch.log 2>&1; echo "batch tests rc=$?"; tail -15 gpurun_out/r03b_batch.log
( time timeout 600 ./tools/mfma_stream_bench ) > gpurun_out/r03b_stream.log 2>&1; echo "stream rc=$?"; grep -E "check|differs|NO " gpurun_out/r03b_stream.log
( time timeout 900 python bench.py ) > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r03b_bench.err; head -c 3000 gpurun_out/r03b_bench.json; echo
( time timeout 600 python -m pytest tests/test_gpu_bench.py -x -q ) > gpurun_out/r03b_benchtests.log 2>&1; echo "bench tests rc=$?"; tail -5 gpurun_out/r03b_benchtests.log
( time timeout 1800 python -m pytest tests -m gpu -q --deselect tests/test_gpu_batch.py --deselect tests/test_gpu_bench.py --deselect tests/test_gpu_round3.py ) > gpurun_out/r03b_suite.log 2>&1; echo "suite rc=$?"; tail -8 gpurun_out/r03b_suite.log
