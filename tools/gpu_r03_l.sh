#!/bin/bash
# round 3, call L: what bounds gemm_stream_kernel at 128 rows (finer ablations)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
L=gpurun_out/r03l_gemmstream.log; : > $L
for shape in "128 4096 4096 1 1" "128 4096 14336 1 1" "128 6144 4096 1 1" "128 4096 4096 2 1" "128 14336 4096 4 2" "4096 4096 4096 4 1"; do
  for d in 0 7 71 15 31 63 127; do timeout 60 tools/gemmstream_bench_$d $shape >> $L 2>&1; done
  echo >> $L
done
cat $L
