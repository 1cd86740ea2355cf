#!/bin/bash
# per-kernel breakdown of the prefill (by kernel and grid) + MFMA busy counters
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
cat > /tmp/agg.py <<'PY'
import csv, collections, sys
d = sys.argv[1]
rows = list(csv.DictReader(open(d + "/t_kernel_trace.csv")))
per = collections.defaultdict(list)
for r in rows:
    nm = r["Kernel_Name"].split("(")[0][-60:]
    per[(nm, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in per.values())
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:10]:
    print("  %-62s grid %5d x %3d  calls %4d  avg %9.1f us  total %8.1f ms  %5.1f%%" % (k[0], k[1], k[2], len(v), sum(v) / len(v), sum(v) / 1e3, 100 * sum(v) / tot))
PY
cd /tmp
rocprofv3 -L 2>/dev/null | grep -io "SQ_[A-Z_]*MFMA[A-Z_0-9]*" | sort -u | head -20
for S in "$@"; do
O=$GRAFT_REPO_ROOT/gpurun_out/prefill_prof/S$S
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python /tmp/pf.py $S > /dev/null 2>&1
echo "S=$S"; python /tmp/agg.py $O
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d ${O}_pmc -o t -- python /tmp/pf.py $S > /dev/null 2> ${O}_pmc.err
python - <<PY
import csv, collections
try:
    rows = list(csv.DictReader(open("${O}_pmc/t_counter_collection.csv")))
except Exception as e:
    print("pmc failed", e); rows = []
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "gemm_mfma" in r["Kernel_Name"]:
        per[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g, c in per.items():
    print("  gemm grid", g, {k: "%.3e" % (sum(v) / len(v)) for k, v in c.items()})
PY
done
