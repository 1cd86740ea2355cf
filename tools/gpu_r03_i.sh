#!/bin/bash
# round 3, call I: gemm_stream_kernel ring depth at one / two batch tiles per wave (short prompts)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
L=gpurun_out/r03i_gemmstream.log; : > $L
for S in 64 128 256; do
  for shape in "6144 4096 1" "4096 4096 1" "4096 14336 1"; do
    set -- $shape
    for b in 0 r44 r65 r65o3 r86; do for ntw in 1 2; do timeout 60 tools/gemmstream_bench_$b $S $1 $2 $ntw $3 >> $L 2>&1; done; done
    echo >> $L
  done
done
cat $L
