#!/bin/bash
# A/B of library variants (tools/build_variant.sh) on the per-kernel decode timings: tools/ab_variants.sh <name> [<name> ...]   ("default" = the in-tree library)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
export TMPDIR=/tmp
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = default ]; then unset LNB_SO; else export LNB_SO=$PWD/llama-nuts-and-bolts_amd/variants/$v.so; fi
  echo "== $v"; timeout 300 python tools/kernel_ab.py ${AB_ITERS:-200} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['us'], d['tok'])"
done; done
