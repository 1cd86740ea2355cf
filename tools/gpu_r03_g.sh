#!/bin/bash
# round 3, call G: gemm_stream_kernel with the hand-issued B-operand reads -- parity, then shape x NTW timing
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
( time timeout 600 python -m pytest tests/test_gpu_batch.py -m gpu -x -q -s -k "prefill" ) > gpurun_out/r03g_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r03g_tests.log
L=gpurun_out/r03g_gemmstream.log; : > $L
for S in 128 256 512 4096; do
  for shape in "6144 4096 1" "4096 4096 1" "14336 4096 2" "4096 14336 1"; do
    set -- $shape
    for b in 0 o33; do for ntw in 1 2; do timeout 60 tools/gemmstream_bench_$b $S $1 $2 $ntw $3 >> $L 2>&1; done; done
    timeout 60 tools/gemmstream_bench_0 $S $1 $2 4 $3 >> $L 2>&1
    for b in 3 7; do timeout 60 tools/gemmstream_bench_$b $S $1 $2 0 $3 >> $L 2>&1; done
    echo >> $L
  done
done
cat $L
