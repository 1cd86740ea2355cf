#!/bin/bash
# round 3, call W: sanity after moving the launch plan into lnb_device.h (same logic): smoke, prefill + batch parity, prefill table
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -1 | cut -c1-200
( timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py -m gpu -x -q ) 2>&1 | tail -2
timeout 300 python tools/prefill_bench.py --modes exact --sizes 128,4096 --stream 2>&1 | cut -c1-170
