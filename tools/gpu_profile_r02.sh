#!/bin/bash
# round-2 profiles: rocprofv3 kernel trace + stats and a FETCH_SIZE pass of bench.py in both modes, the default bench line (with the CPU
# baseline), the configs[2] decode line, the prefill with MFMA counters in both modes, the 70B-like line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
for M in exact fast; do
  L=r02; [ $M = fast ] && L=r02_fast
  O=$PWD/gpurun_out/prof_$L; mkdir -p $O
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --mode $M --steps 32 --warmup 4 --cpu-steps 0 --profile-iters 8 > $O/trace_bench.json 2> $O/trace.err; echo "$M trace rc=$?" )
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o pmc -- python $GRAFT_REPO_ROOT/bench.py --mode $M --steps 8 --warmup 2 --cpu-steps 0 --profile-iters 4 > $O/pmc_fetch_bench.json 2> $O/pmc_fetch.err; echo "$M pmc rc=$?" )
  find $O -name "*.csv" | head; 
done
O=$PWD/gpurun_out/prof_r02
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -4 $O/bench_default.err
timeout 600 python bench.py --mode fast --cpu-steps 0 > gpurun_out/prof_r02_fast/bench_default.json 2>/dev/null
timeout 600 python bench.py --prompt-len 4096 --steps 64 --warmup 8 --cpu-steps 0 > $O/bench_cfg2.json 2>/dev/null; head -c 600 $O/bench_cfg2.json; echo
# configs[2] decode under rocprofv3: the long-context attention kernels per launch, + a FETCH_SIZE pass
O2=$PWD/gpurun_out/prof_r02_cfg2; mkdir -p $O2
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O2/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py --prompt-len 4096 --steps 32 --warmup 4 --cpu-steps 0 --profile-iters 8 > $O2/trace_bench.json 2> $O2/trace.err; echo "cfg2 trace rc=$?" )
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O2/pmc_fetch -o pmc -- python $GRAFT_REPO_ROOT/bench.py --prompt-len 4096 --steps 8 --warmup 2 --cpu-steps 0 --profile-iters 4 > $O2/pmc_fetch_bench.json 2> $O2/pmc_fetch.err; echo "cfg2 pmc rc=$?" )
# prefill: per-kernel times and the matrix-core counters, S = 4096, both modes
cat > /tmp/pf.py <<'PY'
import lnb, sys
S, mode = int(sys.argv[1]), sys.argv[2]
m = lnb.LlamaTransformer(device=0, **lnb.LLAMA_8B).fill_synthetic(1234).finalize(rope_rows=S + 64)
c = lnb.InferenceContext(m, S + 8).set_mode(mode)
toks = lnb.synth_tokens(99, S, 128256)
for _ in range(2):
    c.reset(); _, tok = c.Forward(toks, 0, want_logits=False)
print("tok", tok)
PY
for M in exact fast; do
  P=$PWD/gpurun_out/prof_r02_prefill_$M; mkdir -p $P
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o trace -- python /tmp/pf.py 4096 $M > /dev/null 2> $P/trace.err; echo "prefill $M trace rc=$?" )
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $P/pmc -o pmc -- python /tmp/pf.py 4096 $M > /dev/null 2> $P/pmc.err; echo "prefill $M pmc rc=$?" )
done
python tools/prefill_bench.py --out gpurun_out/prof_r02/prefill.json 2>&1 | tail -9
( time timeout 1200 python bench.py --model llama70b-like --steps 16 --warmup 2 --cpu-steps 0 ) > gpurun_out/prof_r02/bench_70b_like.json 2> gpurun_out/prof_r02/bench_70b_like.err; tail -3 gpurun_out/prof_r02/bench_70b_like.err; head -c 900 gpurun_out/prof_r02/bench_70b_like.json; echo
du -sh gpurun_out
