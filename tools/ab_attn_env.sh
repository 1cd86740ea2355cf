# A/B of an attention switch inside the real configs[2] step (4096-token prompt, 64 decode steps), alternating runs on one box.
# usage: tools/ab_attn_env.sh [ENV_NAME]   (default LNB_ATTN_TOUCH; LNB_ATTN_LAZY was the first use)
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
V=${1:-LNB_ATTN_TOUCH}
for v in 1 0 1 0; do
  env $V=$v timeout 600 python bench.py --prompt-len 4096 --steps 64 --warmup 8 --concurrent 0 --batch-sizes= --cpu-steps 0 --no-traffic-probe --no-configs4 --repeats 3 --profile-iters 16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$V=$v tokens/s', d['value'], 'ms/step', d['ms_per_step'], 'attention us', d['kernels']['attention'], 'golden', d['config']['tokens_vs_oracle_golden'])"
done
