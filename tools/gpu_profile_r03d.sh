#!/bin/bash
# round-3 profiles, final pass (final kernels: streamed prefill, batches up to 128, early-clobber asm loads): rocprofv3 kernel trace + stats and
# FETCH_SIZE of bench.py, the bench lines (default, driver arguments, configs[2], one-GPU pipeline, 70B-like), the 64-sequence batch under rocprofv3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
O=$PWD/gpurun_out/prof_r03; mkdir -p $O
B=$PWD/gpurun_out/prof_r03_batch64; mkdir -p $B
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --cpu-steps 0 --profile-iters 8 --concurrent 0 --batch-sizes "" > $O/trace_bench.json 2> $O/trace.err; echo "trace rc=$?" )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --cpu-steps 0 --profile-iters 4 --concurrent 0 --batch-sizes "" > $O/pmc_fetch_bench.json 2> $O/pmc_fetch.err; echo "pmc rc=$?" )
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -4 $O/bench_default.err
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench_driver_args.json 2> $O/bench_driver_args.err; echo "bench(20) rc=$?"
timeout 900 python bench.py --prompt-len 4096 --steps 64 --warmup 8 --cpu-steps 0 --concurrent 0 --batch-sizes 16 > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "cfg2 rc=$?"; head -c 700 $O/bench_cfg2.json; echo
timeout 300 python tools/batch_bench.py --n 64 --steps 48 --profile-iters 16 > $B/batch_bench.json 2> $B/batch_bench.err; echo "batch rc=$?"; cat $B/batch_bench.json
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $B/trace -o trace -- python $GRAFT_REPO_ROOT/tools/batch_bench.py --n 64 --steps 16 > $B/trace_bench.json 2> $B/trace.err; echo "batch trace rc=$?" )
LNB_FORCE_PIPELINE=1 timeout 600 python bench.py --gpus 1 --steps 64 --warmup 8 --cpu-steps 0 > $O/bench_pipeline_one_gpu.json 2> $O/bench_pipeline_one_gpu.err; echo "pipe rc=$?"; head -c 1800 $O/bench_pipeline_one_gpu.json; echo; tail -3 $O/bench_pipeline_one_gpu.err
( time timeout 1200 python bench.py --model llama70b-like --steps 16 --warmup 2 --cpu-steps 0 ) > $O/bench_70b_like.json 2> $O/bench_70b_like.err; tail -3 $O/bench_70b_like.err; head -c 700 $O/bench_70b_like.json; echo
find $O $B -name "*kernel_trace.csv" -delete; find $O $B -name "*.csv" | head -20; du -sh gpurun_out
