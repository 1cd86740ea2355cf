#!/bin/bash
# round 3, call T: batched attention with the register budget of two workgroups per CU (spills 212 B) -- parity and time
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
( timeout 600 python -m pytest tests/test_gpu_batch.py -m gpu -x -q -k "wide or 8b_shape_equals" ) 2>&1 | tail -2
for o in 0 1; do for n in 64 128; do echo -n "occ4=$o "; LNB_ATTN_BATCH_OCC4=$o timeout 300 python tools/batch_bench.py --n $n --steps 16 --profile-iters 8; done; done 2>&1 | cut -c1-330
