#!/bin/bash
# round 6 GPU runner (one parametrised script; every mode writes under gpurun_out/r06_*).
#   tools/gpu_r06.sh <mode> [args]
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
mkdir -p gpurun_out
mode=$1; shift
ab() {   # per-kernel decode timings of the 8B block shape (tools/kernel_ab.py); env passes through; label = $1
  local label=$1; shift
  echo "== $label"; timeout 300 python tools/kernel_ab.py ${AB_ITERS:-200} 2>&1 | tail -1
}
case "$mode" in
  first)    # first contact: the round's new tests, rung (a) of the FFN ladder, per-kernel baseline
    ( timeout 1500 python -m pytest tests/test_gpu_round6.py -x -q ) 2>&1 | tail -15 | tee gpurun_out/r06_round6_tests.log
    ( timeout 1500 python tools/ffn_overlap.py ) > gpurun_out/r06_ffn_overlap.json 2> gpurun_out/r06_ffn_overlap.err; echo "ffn_overlap rc=$?"; cat gpurun_out/r06_ffn_overlap.json; tail -3 gpurun_out/r06_ffn_overlap.err
    { ab "4 layers"; AB_SCHED=throughput ab "4 layers, throughput forms"; } 2>&1 | tee gpurun_out/r06_ab.log
    ;;
  att)      # decode attention per layer at several contexts, both forms, with the phase stamps of one workgroup
    LNB_GEMV_TIMING=1 timeout 600 python tools/att_timing.py 2>&1 | tee gpurun_out/r06_att_timing.log
    ;;
  prefill)  # exact prefill of the 8B shape at several row counts: the 16-row attention waves against the two-query-tile form
    for v in 0 16 512; do echo "== LNB_ATTN_MFMA2=$v"; LNB_ATTN_MFMA2=$v timeout 900 python tools/prefill_bench.py --sizes 128,256,512,2048,4096 --modes exact 2>&1 | tail -5; done | tee gpurun_out/r06_prefill_attn.log
    ;;
  profile)  # rocprofv3 of the bench command: kernel trace + stats, and the FETCH_SIZE counter pass on its own (tools/summarize_profile.py r06 condenses them)
    O=$PWD/gpurun_out/prof_r06; rm -rf $O; mkdir -p $O
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 4 --repeats 1 --cpu-steps 0 --profile-iters 8 --concurrent 0 --batch-sizes= --no-traffic-probe > $O/trace_bench.json 2> $O/trace.err; echo "trace rc=$?" )
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --repeats 1 --cpu-steps 0 --profile-iters 4 --concurrent 0 --batch-sizes= --no-traffic-probe > $O/pmc_fetch_bench.json 2> $O/pmc_fetch.err; echo "pmc rc=$?" )
    LNB_GEMV_TIMING=1 timeout 300 python tools/kernel_ab.py 50 > gpurun_out/r06_stamps.log 2>&1
    python tools/summarize_profile.py r06 2>&1 | tail -5
    mkdir -p gpurun_out/r06_profiles && cp profiles/r06_rocprofv3_kernel_stats.csv profiles/r06_summary.md profiles/r06_traffic.json gpurun_out/r06_profiles/ 2>/dev/null
    find $O -name "*.csv" | head; du -sh gpurun_out
    ;;
  records)  # the round's other records: the N-GPU code path on one GPU, the matrix-core counters of the exact prefill, batches without the second copy
    LNB_FORCE_PIPELINE=1 LNB_FORCE_PREFLIGHT=1 timeout 900 python bench.py --gpus 1 --steps 64 --warmup 8 --cpu-steps 0 > gpurun_out/r06_bench_pipeline_one_gpu.json 2> gpurun_out/r06_bench_pipeline_one_gpu.err; echo "pipe rc=$?"; tail -3 gpurun_out/r06_bench_pipeline_one_gpu.err
    O=$PWD/gpurun_out/prof_r06_mfma; rm -rf $O; mkdir -p $O
    MD=gpurun_out/r06_prefill_mfma_counters.md; : > $MD
    for S in 128 4096; do
      ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$S -o t_$S -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --modes exact --sizes $S --layers 8 --reps 3 > $O/trace_$S.out 2> $O/trace_$S.err; echo "trace $S rc=$?" )
      ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_$S -o p_$S -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --modes exact --sizes $S --layers 8 --reps 3 > $O/pmc_$S.out 2> $O/pmc_$S.err; echo "pmc $S rc=$?" )
      python tools/mfma_counters.py $O/pmc_$S $O/trace_$S "exact prefill, $S rows, gemm_stream_kernel on the resident weight layouts (no second copy), 8-block cut of the 8B shape" >> $MD
      tail -1 $O/trace_$S.out
    done
    cat $MD | head -40
    timeout 600 python tools/batch_bench.py --n 128 --steps 32 --profile-iters 8 2>&1 | tail -3 | tee gpurun_out/r06_batch_bench.log
    du -sh gpurun_out
    ;;
  tt)       # gemm_stream_kernel's two-weight-tiles-per-wave form (wo / w2 of long prompts): parity test, then the exact prefill with the form off / on (default) / forced
    ( timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -k two_weight_tiles ) 2>&1 | tail -5 | tee gpurun_out/r06_tt.log
    for v in 0 "" 1; do echo "== LNB_GS_TT='$v'"; LNB_GS_TT=$v timeout 900 python tools/prefill_bench.py --sizes 128,512,2048,4096 --modes exact 2>&1 | tail -5; done | tee -a gpurun_out/r06_tt.log
    ;;
  batchsweep)   # batched exact decode: batch tiles per wave forced (LNB_GS_NTW) against the launch rule, with and without the second weight copy
    for n in ${BS_N:-32 64 128}; do for nc in "" "--no-copy"; do for v in "" 1 2 4; do
      LNB_GS_NTW=$v timeout 300 python tools/batch_bench.py --n $n --steps 16 --profile-iters 6 $nc 2>&1 | tail -1
    done; done; done | tee gpurun_out/r06_batch_sweep.log
    ;;
  dvfs)     # are the chain-bound launches clocked by the power budget?  the same kernels on zeroed block matrices (tools/kernel_ab.py AB_ZERO), with the stamps' shader clock
    for z in "" 1; do echo "== AB_ZERO='$z'"; AB_ZERO=$z timeout 600 python tools/kernel_ab.py 200 2>&1 | tail -1; AB_ZERO=$z LNB_GEMV_TIMING=1 timeout 300 python tools/kernel_ab.py 30 2>&1 | grep -E "kernel class|wave 0: n" | head -12; done | tee gpurun_out/r06_dvfs.log
    ;;
  suite)
    ( timeout 2400 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -8 | tee gpurun_out/r06_gpu_suite.log
    ;;
  bench)
    ( time timeout 1200 python bench.py ) > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; echo "bench rc=$?"; tail -3 gpurun_out/r06_bench_default.err
    timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_args.json 2> gpurun_out/r06_bench_driver_args.err; echo "bench(20) rc=$?"
    head -c 600 gpurun_out/r06_bench_default.json; echo
    ;;
  *) echo "unknown mode $mode"; exit 2;;
esac
