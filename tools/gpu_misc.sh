#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( time python bench.py ) > gpurun_out/v_bench.json 2> gpurun_out/v_bench.err; tail -5 gpurun_out/v_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/v_bench.json")); print(d["value"], d["roofline"]["frac"], d["roofline"]["whole_step"]["frac"], d.get("sequences_in_flight"), d["cpu_baseline"]["value"])
PY
python bench.py --mode fast --cpu-steps 0 > gpurun_out/v_bench_fast.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open("gpurun_out/v_bench_fast.json")); print(d["value"], d["roofline"]["whole_step"]["frac"], d.get("sequences_in_flight"))
PY
