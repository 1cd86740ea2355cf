#!/bin/bash
# round 4 GPU runner: one parametrised script instead of one file per call.
#   tools/gpu_r04.sh <mode> [args]   -- modes: linear | ab | suite | bench | ...  (see the case below)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
mkdir -p gpurun_out
mode=$1; shift
case "$mode" in
  linear)   # operator parity of the GEMV kernels + A/B of the per-kernel decode timings
    ( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "linear" ) 2>&1 | tail -3
    timeout 300 python tools/kernel_ab.py 200 2>&1 | tail -1
    LNB_ROWCAST_LDS=0 timeout 300 python tools/kernel_ab.py 200 2>&1 | tail -1
    ;;
  ab)       # per-kernel timings only; env passes through
    timeout 300 python tools/kernel_ab.py ${1:-200} 2>&1 | tail -1
    ;;
  stamps)   # in-kernel cycle stamps of every GEMV class (LNB_GEMV_TIMING), one pass
    LNB_GEMV_TIMING=1 timeout 300 python tools/kernel_ab.py 50 2>&1 | grep -v "^\[timing\]   wave [0-9]: n=0" | tail -${1:-60}
    ;;
  suite)    # whole GPU suite
    ( timeout 3000 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -5
    ;;
  bench)
    timeout 900 python bench.py "$@" 2>&1 | tail -1 | tee gpurun_out/bench_last.json
    ;;
  *) echo "unknown mode $mode"; exit 2;;
esac
