#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_pipeline_cabi.py -q -m gpu -x -k "rccl_exchange" > gpurun_out/u_rccl.log 2>&1; tail -15 gpurun_out/u_rccl.log
timeout 900 python -m pytest tests/test_gpu_fast.py -q -m gpu -x > gpurun_out/u_fast.log 2>&1; tail -3 gpurun_out/u_fast.log
for n in 2 3 4 8; do LNB_FORCE_PIPELINE=1 LNB_PIPELINE_SEQS=$n timeout 400 python bench.py --steps 32 --warmup 4 > gpurun_out/u_pipe_seq$n.json 2> gpurun_out/u_pipe_seq$n.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/u_pipe_seq$n.json")); print($n, d["value"], d["roofline"]["frac"], d["config"].get("host_enqueue_us_per_tick"))
except Exception as e: print($n, "failed", e)
PY
done
