#!/bin/bash
# round 3, last profile refresh with the final kernels: rocprofv3 kernel stats of the 128-sequence batch and of the 4096- / 128-row streamed prefill (+ MFMA counters)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
B=$PWD/gpurun_out/prof_r03_batch128; mkdir -p $B
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $B/trace -o trace -- python $GRAFT_REPO_ROOT/tools/batch_bench.py --n 128 --steps 16 > $B/trace_bench.json 2> $B/trace.err; echo "batch trace rc=$?" )
cat > /tmp/pf.py <<'PY'
import lnb, sys
S = int(sys.argv[1])
m = lnb.LlamaTransformer(device=0, **lnb.LLAMA_8B).fill_synthetic(1234).finalize(rope_rows=S + 64).enable_batch()
c = lnb.InferenceContext(m, S + 8)
toks = lnb.synth_tokens(99, S, 128256)
for _ in range(3):
    c.reset(); _, tok = c.Forward(toks, 0, want_logits=False)
print("tok", tok)
PY
for S in 4096 128; do
  P=$PWD/gpurun_out/prof_r03_prefill2_$S; mkdir -p $P
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o trace -- python /tmp/pf.py $S > $P/trace.out 2> $P/trace.err; echo "prefill $S trace rc=$?" )
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $P/pmc -o pmc -- python /tmp/pf.py $S > $P/pmc.out 2> $P/pmc.err; echo "prefill $S pmc rc=$?" )
  python tools/mfma_counters.py $P/pmc $P/trace "exact prefill, $S rows, streamed (gemm_stream_kernel), final kernels" > $P/summary.md; head -12 $P/summary.md
done
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -size +20M -delete; du -sh gpurun_out
