#!/bin/bash
# round 4 GPU runner: one parametrised script instead of one file per call.
#   tools/gpu_r04.sh <mode> [args]   -- modes: linear | ab | suite | bench | ...  (see the case below)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
mkdir -p gpurun_out
mode=$1; shift
case "$mode" in
  linear)   # operator parity of the GEMV kernels + A/B of the per-kernel decode timings
    ( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "linear or rw24 or rmsnorm" ) 2>&1 | tail -8
    timeout 300 python tools/kernel_ab.py 200 2>&1 | tail -1
    LNB_ROWCAST_LDS=0 timeout 300 python tools/kernel_ab.py 200 2>&1 | tail -1
    ;;
  ab)       # per-kernel timings only; env passes through
    timeout 300 python tools/kernel_ab.py ${1:-200} 2>&1 | tail -1
    ;;
  stamps)   # in-kernel cycle stamps of every GEMV class (LNB_GEMV_TIMING), one pass
    LNB_GEMV_TIMING=1 timeout 300 python tools/kernel_ab.py 50 2>&1 | grep -v "^\[timing\]   wave [0-9]: n=0" | tail -${1:-60}
    ;;
  trace)    # rocprofv3 kernel-trace averages of the decode kernels (true durations, no launch gaps); env passes through
    O=$PWD/gpurun_out/trace_$$; mkdir -p $O
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $GRAFT_REPO_ROOT/tools/kernel_ab.py ${1:-60} > $O/out.json 2> $O/err.log )
    python - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("%-110s calls %6s avg %9.2f us" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
    ;;
  profile)  # the round's record: 8-layer configs[2] oracle golden (host cores, in the background), rocprofv3 trace + FETCH_SIZE pass of the bench, bench lines
    O=$PWD/gpurun_out/prof_r04; mkdir -p $O
    GP=""
    if [ "${LNB_REGEN_GOLDEN:-0}" = 1 ]; then ( timeout 2400 python tests/golden/make_configs2_cut_tokens.py 8 72 $O > $O/golden8.log 2>&1 ) & GP=$!; fi
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --repeats 1 --cpu-steps 0 --profile-iters 8 --concurrent 0 --batch-sizes "" --no-traffic-probe > $O/trace_bench.json 2> $O/trace.err; echo "trace rc=$?" )
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --repeats 1 --cpu-steps 0 --profile-iters 4 --concurrent 0 --batch-sizes "" --no-traffic-probe > $O/pmc_fetch_bench.json 2> $O/pmc_fetch.err; echo "pmc rc=$?" )
    LNB_GEMV_TIMING=1 timeout 300 python tools/kernel_ab.py 50 > $O/stamps.log 2>&1
    timeout 120 tools/chainbench3 > $O/chainbench3.log 2>&1; timeout 60 tools/chainbench4 > $O/chainbench4.log 2>&1
    if [ -n "$GP" ]; then wait $GP; tail -2 $O/golden8.log; [ -f $O/configs2_8layer_tokens.json ] && cp $O/configs2_8layer_tokens.json tests/golden/; fi
    LNB_ATTN_GQA_DBG=1 timeout 300 python tools/gqa_stamps.py > $O/gqa_stamps.log 2>&1
    ( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
    timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_args.json 2> $O/bench_driver_args.err; echo "bench(20) rc=$?"
    timeout 900 python bench.py --prompt-len 4096 --steps 64 --warmup 8 --concurrent 0 --batch-sizes "" > $O/bench_configs2.json 2> $O/bench_configs2.err; echo "cfg2 rc=$?"; head -c 700 $O/bench_configs2.json; echo
    timeout 600 python bench.py --model llama8b-8l --prompt-len 4096 --steps 64 --warmup 4 --cpu-steps 0 --concurrent 0 --batch-sizes "" > $O/bench_configs2_8layer.json 2> $O/bench_configs2_8layer.err; echo "cfg2 8-layer rc=$?"; head -c 900 $O/bench_configs2_8layer.json; echo; tail -2 $O/bench_configs2_8layer.err
    head -c 600 $O/bench_default.json; echo
    find $O -name "*.csv" | head; du -sh gpurun_out
    ;;
  batchprof)  # rocprofv3 kernel stats of the batched step at 16 and 128 sequences (8 layers of the 8B shape: the prompts' prefill dominates a full-depth trace)
    for n in 16 128; do
      O=$PWD/gpurun_out/prof_r04_batch_n$n; mkdir -p $O
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $GRAFT_REPO_ROOT/tools/batch_bench.py --n $n --steps 16 --layers 8 > $O/out.json 2> $O/err.log; echo "n=$n rc=$?" )
      f=$(find $O -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/r04_batch_kernel_stats_n$n.csv
      tail -1 $O/out.json | cut -c1-300
    done
    ;;
  suite)    # whole GPU suite
    ( timeout 3000 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -5
    ;;
  bench)
    timeout 900 python bench.py "$@" 2>&1 | tail -1 | tee gpurun_out/bench_last.json
    ;;
  *) echo "unknown mode $mode"; exit 2;;
esac
