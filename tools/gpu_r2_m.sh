#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
cat > /tmp/pf.py <<'PY'
import lnb, sys
S, mode = int(sys.argv[1]), sys.argv[2]
m = lnb.LlamaTransformer(device=0, **dict(lnb.LLAMA_8B, n_layers=4)).fill_synthetic(1234).finalize(rope_rows=S + 64)
c = lnb.InferenceContext(m, S + 8).set_mode(mode)
toks = lnb.synth_tokens(99, S, 128256)
for _ in range(2):
    c.reset(); _, tok = c.Forward(toks, 0, want_logits=False)
PY
cd /tmp; rm -rf /tmp/gp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d /tmp/gp -o p -- python /tmp/pf.py 4096 fast > /dev/null 2>&1
python - <<'PY'
import csv, collections, glob
f = glob.glob("/tmp/gp/**/p_counter_collection.csv", recursive=True)[0]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "fast_gemm" in r["Kernel_Name"] or "fast_attn" in r["Kernel_Name"]:
        per[(r["Kernel_Name"][-60:], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in per.items():
    g = lambda n: sum(c[n]) / len(c[n]) if c.get(n) else 0
    wc = g("SQ_WAVE_CYCLES")
    print(k, "n=%d" % len(c["SQ_WAVE_CYCLES"]), " ".join("%s=%.3f" % (n.replace("SQ_", ""), g(n) / wc) for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM")), "wave_cycles=%.3e busy=%.3e" % (wc, g("SQ_BUSY_CYCLES")))
PY
