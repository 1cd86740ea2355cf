cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
for v in 1 0 1 0; do
  LNB_ATTN_LAZY=$v timeout 600 python bench.py --prompt-len 4096 --steps 64 --warmup 8 --concurrent 0 --batch-sizes= --cpu-steps 0 --no-traffic-probe --repeats 3 --profile-iters 16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('lazy=$v', d['value'], d['ms_per_step'], d['kernels']['attention'], d['config']['tokens_vs_oracle_golden'])"
done
