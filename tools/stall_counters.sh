# wave-cycle buckets of the exact prefill's kernels (attn_mfma_kernel, gemm_stream_kernel): separate --pmc passes, 4096 rows, 4-block cut of the 8B shape
# usage (on the GPU box): bash tools/stall_counters.sh [rows] [extra env assignments...]
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd TMPDIR=/tmp
S=${1:-4096}; shift
O=$PWD/gpurun_out/stall_$S; rm -rf $O; mkdir -p $O
( cd /tmp && rocprofv3 -L > $O/counters_available.txt 2>&1 )
grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9a-z]*\|TCC_[A-Z_0-9a-z\[\]]*" $O/counters_available.txt | sort -u > $O/counter_names.txt; wc -l $O/counter_names.txt
i=0
while read -r P; do
  i=$((i+1))
  ( cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pass_$i -o p -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --modes exact --sizes $S --layers 4 --reps 2 > $O/pass_$i.out 2> $O/pass_$i.err; echo "pass $i ($P) rc=$?" )
done <<'PASSES'
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES
SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU
SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
SQ_WAVE_CYCLES TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
SQ_WAVE_CYCLES TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
PASSES
python tools/stall_counters.py $O "exact prefill, $S rows, 4-block cut of the 8B shape $*" | tee $O/summary.md
grep -l "rror" $O/pass_*.err 2>/dev/null | head; for f in $O/pass_*.err; do tail -2 $f; done | head -20
du -sh $O; find $O -name "*.csv" -size +20M -delete
