#!/bin/bash
# round 3, call O: what rank 7 of an 8-GPU run holds -- 16 groups x 32 sequences = 512 contexts (streams, events, pinned words) in one process
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd TMPDIR=/tmp
( time LNB_FORCE_PIPELINE=1 LNB_PIPELINE_SEQS=16 LNB_PIPELINE_BATCH=32 timeout 900 python bench.py --gpus 1 --model llama8b-2l --steps 20 --warmup 5 --cpu-steps 0 ) > gpurun_out/r03o_pipe512.json 2> gpurun_out/r03o_pipe512.err
echo "rc=$?"; tail -5 gpurun_out/r03o_pipe512.err; head -c 1500 gpurun_out/r03o_pipe512.json; echo
