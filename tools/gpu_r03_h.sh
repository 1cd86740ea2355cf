#!/bin/bash
# round 3, call H: full GPU suite with the streaming prefill in place, the prefill table (LDS-tiled vs streaming), default bench line
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/ -q -m gpu -x --durations=15 ) > gpurun_out/r03h_suite.log 2>&1
echo "suite rc=$?"; tail -22 gpurun_out/r03h_suite.log
( timeout 500 python tools/prefill_bench.py --modes exact --sizes 16,64,128,256,512,2048,4096 --out gpurun_out/r03h_prefill_tiled.json ) > gpurun_out/r03h_prefill_tiled.log 2>&1
echo "tiled rc=$?"
( timeout 500 python tools/prefill_bench.py --modes exact --sizes 16,64,128,256,512,2048,4096 --stream --out gpurun_out/r03h_prefill_stream.json ) > gpurun_out/r03h_prefill_stream.log 2>&1
echo "stream rc=$?"; cat gpurun_out/r03h_prefill_stream.log | cut -c1-200
( timeout 600 python bench.py ) > gpurun_out/r03h_bench_default.json 2> gpurun_out/r03h_bench_default.err
echo "bench rc=$?"; python - <<'PY'
import json
r = json.load(open("gpurun_out/r03h_bench_default.json"))
print({k: r[k] for k in ("metric", "value", "ms_per_step")}, r["roofline"]["frac"], r["prefill"], r["sequences_in_flight_batched"].get("prefill_streamed"), [(x["n"], x["tokens_per_s"]) for x in r["sequences_in_flight_batched"]["runs"]])
PY
