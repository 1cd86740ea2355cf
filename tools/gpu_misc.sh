#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
run() { tag=$1; shift; ( cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$tag -o t -- python $R/bench.py --steps 8 --warmup 2 --cpu-steps 0 --profile-iters 2 $EXTRA > /tmp/rp_$tag.json 2> /tmp/rp_$tag.err; echo "$tag rc=$? $(head -c 60 /tmp/rp_$tag.json)"; env | grep -c ROCP ); }




python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "concurrent_host_threads" 2>&1 | tail -15
