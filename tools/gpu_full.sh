#!/bin/bash
# full validation: all GPU tests (incl. the 8B 128-token parity run), smoke(), default bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 > gpurun_out/pytest_gpu_full.txt 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu_full.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"; tail -2 gpurun_out/bench_default.err; cat gpurun_out/bench_default.json
