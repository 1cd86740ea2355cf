#!/bin/bash
# round 3: the streamed exact prefill (gemm_stream_kernel) under rocprofv3 -- per-kernel times and matrix-core counters at 4096 and 128 rows
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
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
  P=$PWD/gpurun_out/prof_r03_prefill_$S; mkdir -p $P
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o trace -- python /tmp/pf.py $S > $P/trace.out 2> $P/trace.err; echo "prefill $S trace rc=$?" )
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $P/pmc -o pmc -- python /tmp/pf.py $S > $P/pmc.out 2> $P/pmc.err; echo "prefill $S pmc rc=$?" )
  python tools/mfma_counters.py $P/pmc $P/trace "exact prefill, $S rows, streamed (gemm_stream_kernel)" > $P/summary.md; cat $P/summary.md
  find $P -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03_prefill_${S}_kernel_stats.csv \;
  find $P -name "*kernel_trace.csv" -delete; find $P -name "*counter_collection.csv" -size +20M -delete
done
du -sh gpurun_out
