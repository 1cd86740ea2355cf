#!/bin/bash
# round profile: rocprofv3 kernel trace + stats of bench.py, PMC passes (FETCH_SIZE, SQ counters), full default bench
cd "$GRAFT_REPO_ROOT" || exit 1
R=${1:-r01}
mkdir -p gpurun_out/prof_$R; export TMPDIR=/tmp
export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
O=$PWD/gpurun_out/prof_$R
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --cpu-steps 0 --profile-iters 8 > $O/trace_bench.json 2> $O/trace.err; echo "trace rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --cpu-steps 0 --profile-iters 4 > $O/pmc_fetch_bench.json 2> $O/pmc_fetch.err; echo "pmc fetch rc=$?"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --cpu-steps 0 --profile-iters 4 > $O/pmc_sq_bench.json 2> $O/pmc_sq.err; echo "pmc sq rc=$?"
cd $GRAFT_REPO_ROOT
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -4 $O/bench_default.err; cat $O/bench_default.json | head -c 3000
find $O -name "*.csv" | head -20; du -sh $O
