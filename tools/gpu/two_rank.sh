# usage: bash tools/gpu/two_rank.sh "ENV=..." ...   -- the torchrun line of the driver for 2 ranks on ONE device (gloo), per environment: per-step times in order
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=29620
for e in "$@"; do
  P=$((P+1))
  echo "== $e"
  env $e AO_AMD_BENCH_ONE_DEVICE=1 AO_AMD_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-ops --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); h = d['host']
print(round(d['ms_per_step'], 1), h['step_ms']['in_order'], d['config'].get('all_reduce', {}).get('ms_per_step_median'))"
done
