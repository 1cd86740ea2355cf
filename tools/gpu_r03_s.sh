#!/bin/bash
# round 3, call S: after the dispatch-order / cache-policy change -- batch + prefill parity, prefill table, batched decode sizes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
( timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_full_8b.py -m gpu -x -q -s ) 2>&1 | grep -a "tokens/s\|passed\|failed\|rror" | tail -6
timeout 300 python tools/prefill_bench.py --modes exact --sizes 16,64,128,256,512,2048,4096 --stream --out gpurun_out/r03s_prefill_stream.json 2>&1 | cut -c1-170
for n in 32 64 128; do timeout 300 python tools/batch_bench.py --n $n --steps 32 --profile-iters 8; done 2>&1 | tee gpurun_out/r03s_batch.log
