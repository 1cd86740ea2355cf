#!/bin/bash
# does the configs[2] prefill of the FULL default bench line run the scores-once attention?  (r06: one default run showed 528 ms where every other run shows 495-497)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
show() { python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['prefill']['ms'], d['prefill']['warm_ms'], d['configs2']['prefill']['ms'], d['configs2']['prefill']['warm_ms'], d['configs2']['decode']['tokens_per_s'])"; }
for i in 1 2; do
  python bench.py --cpu-steps 0 2>/dev/null | show "default"
  python bench.py --steps 20 --warmup 5 --cpu-steps 0 2>/dev/null | show "driver-args"
done
