#!/bin/bash
# round 4 GPU runner: one parametrised script instead of one file per call.
#   tools/gpu_r04.sh <mode> [args]   -- modes: linear | ab | suite | bench | ...  (see the case below)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
mkdir -p gpurun_out
mode=$1; shift
case "$mode" in
  linear)   # operator parity of the GEMV kernels + A/B of the per-kernel decode timings
    ( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "linear or rw24 or rmsnorm" ) 2>&1 | tail -8
    timeout 300 python tools/kernel_ab.py 200 2>&1 | tail -1
    LNB_ROWCAST_LDS=0 timeout 300 python tools/kernel_ab.py 200 2>&1 | tail -1
    ;;
  ab)       # per-kernel timings only; env passes through
    timeout 300 python tools/kernel_ab.py ${1:-200} 2>&1 | tail -1
    ;;
  stamps)   # in-kernel cycle stamps of every GEMV class (LNB_GEMV_TIMING), one pass
    LNB_GEMV_TIMING=1 timeout 300 python tools/kernel_ab.py 50 2>&1 | grep -v "^\[timing\]   wave [0-9]: n=0" | tail -${1:-60}
    ;;
  trace)    # rocprofv3 kernel-trace averages of the decode kernels (true durations, no launch gaps); env passes through
    O=$PWD/gpurun_out/trace_$$; mkdir -p $O
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $GRAFT_REPO_ROOT/tools/kernel_ab.py ${1:-60} > $O/out.json 2> $O/err.log )
    python - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("%-110s calls %6s avg %9.2f us" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
    ;;
  suite)    # whole GPU suite
    ( timeout 3000 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -5
    ;;
  bench)
    timeout 900 python bench.py "$@" 2>&1 | tail -1 | tee gpurun_out/bench_last.json
    ;;
  *) echo "unknown mode $mode"; exit 2;;
esac
