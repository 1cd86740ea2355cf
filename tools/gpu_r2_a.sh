#!/bin/bash
# round 2, GPU call A: the new exact-mode parity tests, the tolerance-mode tests, both bench modes, the fast-kernel row-group sweep
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -m pytest tests/test_gpu_configs.py -q -m gpu -x > gpurun_out/a_configs.log 2>&1; tail -15 gpurun_out/a_configs.log
python -m pytest tests/test_gpu_fast.py -q -m gpu > gpurun_out/a_fast.log 2>&1; tail -30 gpurun_out/a_fast.log
python bench.py --steps 64 --warmup 8 --cpu-steps 0 > gpurun_out/a_bench_exact.json 2> gpurun_out/a_bench_exact.err; tail -c 1500 gpurun_out/a_bench_exact.json; tail -3 gpurun_out/a_bench_exact.err
python bench.py --mode fast --steps 64 --warmup 8 --cpu-steps 0 > gpurun_out/a_bench_fast.json 2> gpurun_out/a_bench_fast.err; cat gpurun_out/a_bench_fast.json; tail -3 gpurun_out/a_bench_fast.err
for rg in 8 16 32; do LNB_FAST_RG=$rg python bench.py --mode fast --steps 32 --warmup 4 --cpu-steps 0 > gpurun_out/a_bench_fast_rg$rg.json 2>/dev/null; python - <<PY
import json; d=json.load(open("gpurun_out/a_bench_fast_rg$rg.json")); print("RG $rg", d["value"], {k:v["ms"] for k,v in d["kernels"].items()})
PY
done
python tools/fast_mode_stats.py --seeds 4 --tokens 64 --out gpurun_out/a_fast_stats.json
