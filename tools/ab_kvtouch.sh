# A/B of the KV touch (LNB_KV_TOUCH) inside the real decode steps: configs[2] (T ~ 4100) and the headline (T ~ 270)
cd ${GRAFT_REPO_ROOT:-.}; export PYTHONPATH=$PWD:$PWD/llama-nuts-and-bolts_amd
for v in 1 0 1 0; do
  LNB_KV_TOUCH=$v timeout 600 python bench.py --prompt-len 4096 --steps 64 --warmup 8 --concurrent 0 --batch-sizes= --cpu-steps 0 --no-traffic-probe --repeats 3 --profile-iters 16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('configs2 touch=$v', d['value'], d['ms_per_step'], {k: round(v['ms']*1e3,2) for k,v in d['kernels'].items()}, d['config']['tokens_vs_oracle_golden']['identical_prefix'])"
done
for v in 1 0 1 0; do
  LNB_KV_TOUCH=$v timeout 600 python bench.py --steps 256 --warmup 16 --concurrent 0 --batch-sizes= --cpu-steps 0 --no-traffic-probe --no-configs2 --repeats 3 --profile-iters 16 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('headline touch=$v', d['value'], d['ms_per_step'], {k: round(v['ms']*1e3,2) for k,v in d['kernels'].items()}, d['config']['tokens_vs_oracle_golden']['identical_prefix'])"
done
