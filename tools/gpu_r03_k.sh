#!/bin/bash
# round 3, call K: batches of more than 16 sequences (rows of the streaming product) -- parity, pipeline, bench line
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd TMPDIR=/tmp
( time timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_pipeline_cabi.py -q -m gpu -x -s ) > gpurun_out/r03k_tests.log 2>&1
echo "tests rc=$?"; grep -a "tokens/s\|passed\|failed\|Error\|error" gpurun_out/r03k_tests.log | tail -12
( timeout 900 python bench.py ) > gpurun_out/r03k_bench_default.json 2> gpurun_out/r03k_bench_default.err
echo "bench rc=$?"; tail -3 gpurun_out/r03k_bench_default.err; python - <<'PY'
import json
r = json.load(open("gpurun_out/r03k_bench_default.json"))
print({k: r[k] for k in ("metric", "value", "ms_per_step")}, r["roofline"]["frac"], r["prefill"]["ms"])
b = r["sequences_in_flight_batched"]
print(b.get("prefill_streamed"))
for x in b["runs"]:
    print(x["n"], x["tokens_per_s"], x["ms_per_step"], x["sequence0_tokens_vs_single_run"], x.get("kernels_us"))
PY
