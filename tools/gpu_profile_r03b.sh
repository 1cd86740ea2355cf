#!/bin/bash
# round-3 profiles, second pass (the first one, tools/gpu_profile_r03.sh, gave the configs[2] / 70B-like / driver-argument lines): rocprofv3
# kernel trace + stats and FETCH_SIZE of bench.py and of the batched decode, the default bench line, the one-GPU pipeline line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
O=$PWD/gpurun_out/prof_r03; mkdir -p $O
B=$PWD/gpurun_out/prof_r03_batch; mkdir -p $B
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --cpu-steps 0 --profile-iters 8 --concurrent 0 --batch-sizes "" > $O/trace_bench.json 2> $O/trace.err; echo "trace rc=$?" )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --cpu-steps 0 --profile-iters 4 --concurrent 0 --batch-sizes "" > $O/pmc_fetch_bench.json 2> $O/pmc_fetch.err; echo "pmc rc=$?" )
( time timeout 600 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -4 $O/bench_default.err
timeout 300 python tools/batch_bench.py --n 16 --steps 48 --profile-iters 16 > $B/batch_bench.json 2> $B/batch_bench.err; echo "batch rc=$?"; cat $B/batch_bench.json
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $B/trace -o trace -- python $GRAFT_REPO_ROOT/tools/batch_bench.py --n 16 --steps 16 > $B/trace_bench.json 2> $B/trace.err; echo "batch trace rc=$?" )
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $B/pmc_fetch -o pmc -- python $GRAFT_REPO_ROOT/tools/batch_bench.py --n 16 --steps 4 > $B/pmc_fetch_bench.json 2> $B/pmc_fetch.err; echo "batch pmc rc=$?" )
LNB_FORCE_PIPELINE=1 timeout 600 python bench.py --gpus 1 --steps 64 --warmup 8 --cpu-steps 0 > $O/bench_pipeline_one_gpu.json 2> $O/bench_pipeline_one_gpu.err; echo "pipe rc=$?"; head -c 2500 $O/bench_pipeline_one_gpu.json; echo
find $O $B -name "*.csv" | head -20; du -sh gpurun_out
