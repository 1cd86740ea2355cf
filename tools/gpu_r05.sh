#!/bin/bash
# round 5 GPU runner (one parametrised script; every mode writes under gpurun_out/r05_*).
#   tools/gpu_r05.sh <mode> [args]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
mkdir -p gpurun_out
mode=$1; shift
ab() {   # per-kernel decode timings of the 8B block shape (tools/kernel_ab.py); env passes through; label = $1
  local label=$1; shift
  echo "== $label"; timeout 300 python tools/kernel_ab.py ${AB_ITERS:-200} 2>&1 | tail -1
}
case "$mode" in
  first)    # first contact of the round: the new tests, the Infinity-Cache warm-up A/B, the default bench line
    ( timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_bench.py -x -q ) 2>&1 | tail -15
    {
      ab "default (no warm-up)"
      LNB_MALL_EVERY=4 LNB_MALL_WO_UNITS=0 ab "every 4, attention rows only (58 MB)"
      LNB_MALL_EVERY=4 LNB_MALL_ATTN_UNITS=3584 ab "every 4, half attention / half wo"
      LNB_MALL_EVERY=4 LNB_MALL_ATTN_UNITS=0 ab "every 4, wo chain waves only"
      LNB_MALL_EVERY=2 LNB_MALL_ATTN_UNITS=5000 ab "every 2 (117 MB), 5000 units attention / rest wo"
      LNB_MALL_EVERY=8 LNB_MALL_WO_UNITS=0 ab "every 8, attention only (29 MB)"
      LNB_MALL_EVERY=1 LNB_MALL_ATTN_UNITS=6000 LNB_MALL_WO_UNITS=12000 ab "every 1, 47 MB attention + 94 MB wo (prefix of the matrix)"
    } 2>&1 | tee gpurun_out/r05_mall_ab.log
    ( time timeout 900 python bench.py ) > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/r05_bench_default.err
    head -c 1500 gpurun_out/r05_bench_default.json; echo
    ;;
  second)   # the full-depth configs[2] golden on the box's host cores (background, ~25 min of 64 threads) while the GPU runs the suite and the A/Bs
    mkdir -p gpurun_out/gold32
    ( timeout 3000 python tests/golden/make_configs2_cut_tokens.py 32 100 gpurun_out/gold32 > gpurun_out/gold32/log.txt 2>&1 ) & GP=$!
    ( timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -8 | tee gpurun_out/r05_gpu_suite.log
    {
      ab "4 layers"
      AB_LAYERS=32 ab "32 layers"
      AB_SCHED=throughput ab "4 layers, throughput forms"
    } 2>&1 | tee gpurun_out/r05_ab_layers.log
    Q="--steps 48 --warmup 4 --batch-sizes= --cpu-steps 0 --no-traffic-probe --no-configs2 --repeats 1 --profile-iters 4"
    for v in "LNB_TP_W13=1" "LNB_TP_W13=0"; do
      echo "== concurrent, $v"; env $v timeout 300 python bench.py $Q 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], json.dumps(d['sequences_in_flight']))"
    done 2>&1 | tee gpurun_out/r05_concurrent_ab.log
    ( time timeout 900 python bench.py ) > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/r05_bench_default.err
    timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_args.json 2> gpurun_out/r05_bench_driver_args.err; echo "bench(20) rc=$?"
    wait $GP; tail -2 gpurun_out/gold32/log.txt; ls -la gpurun_out/gold32
    ;;
  third)    # the round's records on an otherwise idle box: new tests, default line, driver arguments, configs[2], 70B-like shape, the N-GPU code path on one GPU
    ( timeout 900 python -m pytest tests/test_gpu_round5.py -x -q ) 2>&1 | tail -6 | tee gpurun_out/r05_round5_tests.log
    ( time timeout 900 python bench.py ) > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/r05_bench_default.err
    timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_args.json 2> gpurun_out/r05_bench_driver_args.err; echo "bench(20) rc=$?"
    timeout 900 python bench.py --prompt-len 4096 --steps 64 --warmup 8 --concurrent 0 --batch-sizes= --cpu-steps 0 > gpurun_out/r05_bench_configs2.json 2> gpurun_out/r05_bench_configs2.err; echo "cfg2 rc=$?"; tail -2 gpurun_out/r05_bench_configs2.err
    ( time timeout 1200 python bench.py --model llama70b-like --steps 16 --warmup 2 --cpu-steps 0 ) > gpurun_out/r05_bench_70b_like.json 2> gpurun_out/r05_bench_70b_like.err; echo "70b rc=$?"; tail -3 gpurun_out/r05_bench_70b_like.err
    LNB_FORCE_PIPELINE=1 LNB_FORCE_PREFLIGHT=1 timeout 900 python bench.py --gpus 1 --steps 64 --warmup 8 --cpu-steps 0 > gpurun_out/r05_bench_pipeline_one_gpu.json 2> gpurun_out/r05_bench_pipeline_one_gpu.err; echo "pipe rc=$?"; tail -4 gpurun_out/r05_bench_pipeline_one_gpu.err
    for f in default driver_args configs2 70b_like pipeline_one_gpu; do echo "--- $f"; head -c 400 gpurun_out/r05_bench_$f.json; echo; done
    ;;
  queues)   # sequences in flight against the number of hardware queues the HIP runtime spreads the streams over (GPU_MAX_HW_QUEUES, default 4)
    Q="--steps 48 --warmup 4 --batch-sizes= --cpu-steps 0 --no-traffic-probe --no-configs2 --repeats 1 --profile-iters 4"
    for q in ${QUEUES:-default 2 8 16}; do
      for n in ${NSEQ:-2 8 16}; do
        if [ "$q" = default ]; then E="LNB_DUMMY=1"; else E="GPU_MAX_HW_QUEUES=$q"; fi
        echo "== GPU_MAX_HW_QUEUES=$q, $n sequences in flight"
        env $E timeout 300 python bench.py $Q --concurrent $n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['sequences_in_flight']
print('single', d['value'], '| n', s['n'], 'throughput', s['tokens_per_s'], 'latency', s['latency_forms_same_run']['tokens_per_s'], '| n2', s.get('n2',{}).get('tokens_per_s'), s.get('n2',{}).get('latency_forms_same_run',{}).get('tokens_per_s'))"
      done
    done 2>&1 | tee gpurun_out/r05_hw_queues.log
    ;;
  handoff)  # kernel boundary against an in-kernel grid hand-off on the decode kernels' launch shape (tools/handoff_bench.hip)
    [ -x tools/handoff_bench ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/handoff_bench.hip -o tools/handoff_bench
    for steps in 2048 4096 8192; do timeout 120 tools/handoff_bench $steps; done 2>&1 | tee gpurun_out/r05_handoff_bench.log
    ;;
  variants) # compile-time variants of the library (ab_variants/*.so, built by hand with -D switches) against the default build, per-kernel timings
    { ab "default build"
      for so in ab_variants/*.so; do LNB_SO=$PWD/$so ab "$(basename $so .so)"; done
      ab "default build again"; } 2>&1 | tee gpurun_out/r05_variants.log
    ;;
  profile)  # rocprofv3 of the bench command: kernel trace + stats, and the FETCH_SIZE counter pass on its own (tools/summarize_profile.py r05 condenses them)
    O=$PWD/gpurun_out/prof_r05; mkdir -p $O
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --repeats 1 --cpu-steps 0 --profile-iters 8 --concurrent 0 --batch-sizes= --no-traffic-probe > $O/trace_bench.json 2> $O/trace.err; echo "trace rc=$?" )
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --repeats 1 --cpu-steps 0 --profile-iters 4 --concurrent 0 --batch-sizes= --no-traffic-probe > $O/pmc_fetch_bench.json 2> $O/pmc_fetch.err; echo "pmc rc=$?" )
    LNB_GEMV_TIMING=1 timeout 300 python tools/kernel_ab.py 50 > gpurun_out/r05_stamps.log 2>&1
    find $O -name "*.csv" | head; du -sh gpurun_out
    ;;
  final)    # the round's closing record on one box: whole GPU suite, then the bench lines
    ( timeout 1500 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -6 | tee gpurun_out/r05_gpu_suite.log
    $0 third
    ;;
  prefill)  # exact prefill: the streaming kernel on the RESIDENT layouts (default) against the LDS-tiled kernel and the M16 copy; parity tests first
    ( timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_full_8b.py tests/test_gpu_round5.py -x -q -k "prefill or golden or 8b or full" ) 2>&1 | tail -5
    ( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q ) 2>&1 | tail -3
    { echo "== resident layouts (default)"; timeout 600 python tools/prefill_bench.py --modes exact --sizes 16,64,128,256,512,2048,4096
      echo "== LDS-tiled kernel (LNB_PREFILL_NATIVE=0)"; LNB_PREFILL_NATIVE=0 timeout 600 python tools/prefill_bench.py --modes exact --sizes 16,64,128,256,512,2048,4096
      echo "== M16 copy (--stream)"; timeout 600 python tools/prefill_bench.py --modes exact --stream --sizes 16,64,128,256,512,2048,4096; } 2>&1 | tee gpurun_out/r05_prefill.log
    ;;
  mfma)     # matrix-core counters of the exact prefill on the RESIDENT layouts (the round's default prompt path): a trace pass and a counter pass per size, 8-block cut
    O=$PWD/gpurun_out/prof_r05_mfma; rm -rf $O; mkdir -p $O
    MD=gpurun_out/r05_prefill_mfma_counters.md; : > $MD
    for S in 128 4096; do
      ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$S -o t_$S -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --modes exact --sizes $S --layers 8 --reps 3 > $O/trace_$S.out 2> $O/trace_$S.err; echo "trace $S rc=$?" )
      ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_$S -o p_$S -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --modes exact --sizes $S --layers 8 --reps 3 > $O/pmc_$S.out 2> $O/pmc_$S.err; echo "pmc $S rc=$?" )
      python tools/mfma_counters.py $O/pmc_$S $O/trace_$S "exact prefill, $S rows, gemm_stream_kernel on the resident weight layouts (no second copy), 8-block cut of the 8B shape" >> $MD
      tail -1 $O/trace_$S.out
    done
    cat $MD; find $O -name "*kernel_stats.csv" -exec cp {} gpurun_out/ \; ; du -sh gpurun_out
    ;;
  closing)  # after the last kernel change: quick parity, the throughput forms, the default / driver lines, the rocprofv3 record
    ( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py -x -q ) 2>&1 | tail -3
    ( time timeout 900 python bench.py ) > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/r05_bench_default.err
    timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_driver_args.json 2> gpurun_out/r05_bench_driver_args.err; echo "bench(20) rc=$?"
    $0 profile
    ;;
  ab)       # env passes through
    ab "${1:-custom}"
    ;;
  suite)    # whole GPU suite
    ( timeout 3000 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -8 | tee gpurun_out/r05_gpu_suite.log
    ;;
  bench)
    timeout 900 python bench.py "$@" > gpurun_out/r05_bench_last.json 2> gpurun_out/r05_bench_last.err; echo "rc=$?"; head -c 600 gpurun_out/r05_bench_last.json; echo; tail -3 gpurun_out/r05_bench_last.err
    ;;
  *) echo "unknown mode $mode"; exit 2;;
esac
