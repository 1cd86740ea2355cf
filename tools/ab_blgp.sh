# A/B of gemm_blgp_kernel (LNB_GEMM_BLGP) for short prompts: parity suites first, then the exact prefill at 16 .. 512 rows
cd ${GRAFT_REPO_ROOT:-.}; export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
( timeout 1500 python -m pytest tests/test_gpu_batch.py tests/test_gpu_parity.py tests/test_gpu_full_8b.py -x -q -k "prefill or gemm or 8b or full or linear" ) 2>&1 | tail -4
for v in 0 1 2; do echo "== LNB_GEMM_BLGP=$v"; LNB_GEMM_BLGP=$v timeout 600 python tools/prefill_bench.py --modes exact --sizes 16,32,64,128,256,512 2>&1 | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('  rows',d['rows'],'ms',d['ms'],'tok',d['next_token'])"; done
