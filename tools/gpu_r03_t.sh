#!/bin/bash
# round 3, call T2: batched attention compiled for two workgroups per CU (no K ping-pong: 120 VGPRs) -- parity and time
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
( timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_pipeline_cabi.py -m gpu -x -q ) 2>&1 | tail -2
for o in 0 1; do for n in 32 64 128; do echo -n "dense=$o "; LNB_ATTN_BATCH_DENSE=$o timeout 300 python tools/batch_bench.py --n $n --steps 16 --profile-iters 8; done; done 2>&1 | cut -c1-330
