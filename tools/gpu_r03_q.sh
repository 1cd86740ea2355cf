#!/bin/bash
# round 3, call Q2: dispatch order of gemm_stream_kernel (GS_ORDER 0 = tile groups fastest, 1 = row groups fastest) by row count
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
L=gpurun_out/r03q_gemmstream.log; : > $L
for S in 256 512 1024 2048 4096; do for shape in "6144 4096 0 1" "4096 4096 0 1" "14336 4096 0 2" "4096 14336 0 1"; do for o in 0 1; do echo -n "order=$o " >> $L; GS_ORDER=$o timeout 60 tools/gemmstream_bench_0 $S $shape >> $L 2>&1; done; done; done
sed 's/ lds=[0-9]*//; s/ err=no error//; s/occ=2\/2 R=3\/3 //; s/GS_DBG=0 //; s/ per launch//' $L
( timeout 600 python -m pytest tests/test_gpu_batch.py -m gpu -x -q -k "prefill or wide" ) 2>&1 | tail -2
timeout 300 python tools/prefill_bench.py --modes exact --sizes 128,512,1024,2048,4096 --stream 2>&1 | cut -c1-170
