#!/bin/bash
# round 3, call U: pipeline bench with groups of 64 sequences: one-GPU form (2 groups) and what rank 7 of 8 holds (16 groups = 1024 contexts, two-layer cut)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
( time LNB_FORCE_PIPELINE=1 timeout 900 python bench.py --gpus 1 --steps 64 --warmup 8 --cpu-steps 0 ) > gpurun_out/r03u_pipe_one_gpu.json 2> gpurun_out/r03u_pipe_one_gpu.err; echo "rc=$?"; tail -4 gpurun_out/r03u_pipe_one_gpu.err; head -c 600 gpurun_out/r03u_pipe_one_gpu.json; echo
( time LNB_FORCE_PIPELINE=1 LNB_PIPELINE_SEQS=16 timeout 900 python bench.py --gpus 1 --model llama8b-2l --steps 20 --warmup 5 --cpu-steps 0 ) > gpurun_out/r03u_pipe1024.json 2> gpurun_out/r03u_pipe1024.err; echo "rc=$?"; tail -4 gpurun_out/r03u_pipe1024.err; head -c 400 gpurun_out/r03u_pipe1024.json; echo
