cd ${GRAFT_REPO_ROOT:-.}; export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd TMPDIR=/tmp
O=$PWD/gpurun_out/prof_blgp; rm -rf $O; mkdir -p $O
for S in 32 128; do
  ( cd /tmp && LNB_GEMM_BLGP=2 timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$S -o t -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --modes exact --sizes $S --layers 8 --reps 3 > $O/t_$S.out 2> $O/t_$S.err )
  echo "== S=$S BLGP=2"; python - <<PY
import csv,glob
f=glob.glob("$O/t_$S/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'gemm' in n or 'attn' in n: print('  %-60s calls %s avg %.1f us' % (n[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
done
