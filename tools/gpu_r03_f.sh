#!/bin/bash
# round 3, call F: gemm_stream_kernel -- parity, then shape x NTW timing and ablations (tools/gemmstream_bench.hip)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
( time timeout 600 python -m pytest tests/test_gpu_batch.py -m gpu -x -q -s -k "prefill" ) > gpurun_out/r03f_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r03f_tests.log
L=gpurun_out/r03f_gemmstream.log; : > $L
timeout 60 tools/gemmstream_bench_0 peak >> $L 2>&1
for S in 4096 512 128; do
  for shape in "6144 4096 1" "4096 4096 1" "14336 4096 2" "4096 14336 1"; do
    set -- $shape
    for ntw in 1 2 4 8; do timeout 60 tools/gemmstream_bench_0 $S $1 $2 $ntw $3 >> $L 2>&1; done
    for d in 1 3 7; do timeout 60 tools/gemmstream_bench_$d $S $1 $2 4 $3 >> $L 2>&1; done
    echo >> $L
  done
done
cat $L
