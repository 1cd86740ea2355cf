#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python __graft_entry__.py smoke ) > gpurun_out/t_warm.log 2>&1
timeout 200 python tools/kernel_ab.py >> gpurun_out/t_warm.log 2>&1
for f in .variants/*.so; do LNB_SO=$PWD/$f timeout 200 python tools/kernel_ab.py >> gpurun_out/t_warm.log 2>&1; done
LNB_GEMV_TIMING=1 timeout 200 python tools/kernel_ab.py 20 2>&1 | grep -A8 "class 3" | tail -9 | grep -E "wave [037]" >> gpurun_out/t_warm.log 2>&1
tail -30 gpurun_out/t_warm.log
