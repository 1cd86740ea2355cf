#!/bin/bash
# round 3, call V: one-GPU pipeline path with the final kernels: 2 / 3 / 4 batches of 64 in flight (and 2 x 128)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
for cfg in "2 64" "3 64" "4 64" "2 128"; do set -- $cfg
  LNB_FORCE_PIPELINE=1 LNB_PIPELINE_SEQS=$1 LNB_PIPELINE_BATCH=$2 timeout 600 python bench.py --gpus 1 --steps 64 --warmup 8 --cpu-steps 0 > gpurun_out/r03v_pipe_$1x$2.json 2> gpurun_out/r03v_pipe_$1x$2.err
  python - <<PY
import json
r = json.loads(open("gpurun_out/r03v_pipe_$1x$2.json").read().strip().splitlines()[-1]); b = r["config"]["batched"]
print("groups $1 x batch $2:", b["tokens_per_s"], "tokens/s,", b["sequences_in_flight"], "in flight, golden", b["tokens_vs_oracle_golden"]["identical_prefix"], "/", b["tokens_vs_oracle_golden"]["compared"], "host us/tick", b["host_enqueue_us_per_tick"])
PY
done 2>&1 | tee gpurun_out/r03v_pipe.log
