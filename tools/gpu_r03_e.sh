#!/bin/bash
# round 3, call E: the streaming prefill GEMM -- parity first, then the timing table with / without it
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
( time timeout 900 python -m pytest tests/test_gpu_batch.py -m gpu -x -q -s -k "prefill" ) > gpurun_out/r03e_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r03e_tests.log
( timeout 500 python tools/prefill_bench.py --modes exact --sizes 16,64,128,512,2048,4096 --out gpurun_out/r03e_prefill_tiled.json ) > gpurun_out/r03e_prefill_tiled.log 2>&1
echo "tiled rc=$?"; tail -8 gpurun_out/r03e_prefill_tiled.log
( timeout 500 python tools/prefill_bench.py --modes exact --sizes 16,64,128,512,2048,4096 --stream --out gpurun_out/r03e_prefill_stream.json ) > gpurun_out/r03e_prefill_stream.log 2>&1
echo "stream rc=$?"; tail -8 gpurun_out/r03e_prefill_stream.log
