#!/bin/bash
# round 3: what the driver runs at round end, on one fresh box -- the whole GPU suite, smoke(), the default bench line
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests/ -q -m gpu -x --durations=12 ) > gpurun_out/r03_final_suite.log 2>&1; echo "suite rc=$?"; tail -20 gpurun_out/r03_final_suite.log
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r03_final_smoke.log 2>&1; echo "smoke rc=$?"; tail -4 gpurun_out/r03_final_smoke.log
( time timeout 900 python bench.py ) > gpurun_out/r03_final_bench.json 2> gpurun_out/r03_final_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r03_final_bench.err; head -c 400 gpurun_out/r03_final_bench.json; echo
