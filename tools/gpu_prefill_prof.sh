#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
mkdir -p gpurun_out/prefill_prof; export TMPDIR=/tmp
cat > /tmp/pf.py <<'PY'
import lnb, sys
S = int(sys.argv[1])
m = lnb.LlamaTransformer(device=0, **lnb.LLAMA_8B).fill_synthetic(1234).finalize()
c = lnb.InferenceContext(m, S + 8)
toks = lnb.synth_tokens(99, S, 128256)
for _ in range(2):
    c.reset(); _, tok = c.Forward(toks, 0, want_logits=False)
print("tok", tok)
PY
cd /tmp
for S in 128 512; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prefill_prof/S$S -o t -- python /tmp/pf.py $S > /dev/null 2>&1
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/prefill_prof/S$S/t_kernel_stats.csv")))
print("S=$S")
for r in rows[:9]:
    print("  %-90s calls %5s  avg %9.1f us  total %8.1f ms  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
done
