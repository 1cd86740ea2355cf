#!/bin/bash
# round 3, call J: rows norm on the parallel-exact evaluation -- parity, then the prefill table
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_full_8b.py tests/test_gpu_configs.py -q -m gpu -x ) > gpurun_out/r03j_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r03j_tests.log
( timeout 500 python tools/prefill_bench.py --modes exact --sizes 16,64,128,256,512,2048,4096 --stream --out gpurun_out/r03j_prefill_stream.json ) > gpurun_out/r03j_prefill_stream.log 2>&1
echo "stream rc=$?"; cut -c1-175 gpurun_out/r03j_prefill_stream.log
LNB_NORM_ROWS_WIDE=0 timeout 300 python tools/prefill_bench.py --modes exact --sizes 128,4096 --stream 2>&1 | cut -c1-175
