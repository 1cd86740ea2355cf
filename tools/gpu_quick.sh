#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/pytest_parity.txt 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_parity.txt
timeout 600 python bench.py --steps 64 --warmup 8 --cpu-steps 0 > gpurun_out/bench2.json 2> gpurun_out/bench2.err; echo "bench rc=$?"; tail -3 gpurun_out/bench2.err; python -c "
import json; d=json.load(open('gpurun_out/bench2.json')); print(d['value'], d['ms_per_step']); [print(k, v) for k,v in d['kernels'].items()]"
