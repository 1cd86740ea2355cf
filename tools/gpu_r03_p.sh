#!/bin/bash
# round 3, call P: HBM read bytes (FETCH_SIZE) of the streamed prefill's kernels at 4096 and 128 rows; smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -2
cat > /tmp/pf.py <<'PY'
import lnb, sys
S = int(sys.argv[1])
m = lnb.LlamaTransformer(device=0, **dict(lnb.LLAMA_8B, n_layers=4)).fill_synthetic(1234).finalize(rope_rows=S + 64).enable_batch()
c = lnb.InferenceContext(m, S + 8)
toks = lnb.synth_tokens(99, S, 128256)
for _ in range(2):
    c.reset(); _, tok = c.Forward(toks, 0, want_logits=False)
print("tok", tok)
PY
for S in 4096 128; do
  P=$PWD/gpurun_out/prof_r03_prefill_fetch_$S; mkdir -p $P
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $P/pmc -o pmc -- python /tmp/pf.py $S > $P/pmc.out 2> $P/pmc.err; echo "prefill $S pmc rc=$?" )
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$P/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[(r["Kernel_Name"].split("(")[0][:60], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print("S=$S  %-62s grid %-9s launches %3d  HBM read per launch %9.1f MB (FETCH_SIZE KiB x 2 / 1024)" % (k[0], k[1], len(v), 2 * sum(v) / len(v) / 1024.0))
PY
done 2>&1 | tee gpurun_out/r03p_prefill_fetch.log
