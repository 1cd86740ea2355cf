#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/l_bench_driver.json 2> gpurun_out/l_bench_driver.err; echo "rc=$?"; head -c 700 gpurun_out/l_bench_driver.json; echo; tail -2 gpurun_out/l_bench_driver.err
