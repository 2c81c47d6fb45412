# usage: bash tools/gpu/r05_env_ab.sh OUT reps "ENV_A" "ENV_B" ["ENV_C" ...]  -- alternating bench runs under different environments (same library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; R=$2; shift 2; mkdir -p $O
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ops --no-roofline --no-reference-loop > /dev/null 2>&1
for i in $(seq 1 $R); do
  for e in "$@"; do
    env $e python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ops --no-roofline --no-reference-loop 2>$O/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', d['ms_per_step'], d['host']['step_ms'])" || tail -5 $O/err.txt
  done
done 2>&1 | tee $O/env_ab.txt
