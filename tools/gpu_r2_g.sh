#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
for r in 8 12 16; do LNB_RING_R=$r python bench.py --steps 48 --warmup 8 --cpu-steps 0 > gpurun_out/g_ring$r.json 2> gpurun_out/g_ring$r.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/g_ring$r.json")); print("R $r", d["value"], d["config"]["tokens_vs_oracle_golden"]["identical_prefix"], {k:v["ms"] for k,v in d["kernels"].items()})
except Exception as e: print("R $r failed", e, open("gpurun_out/g_ring$r.err").read()[-500:])
PY
done
