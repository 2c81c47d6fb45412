cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in $(seq 1 40); do AO_AMD_BENCH_ONE_DEVICE=1 AO_AMD_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 3 --warmup 2 --points 20000 --no-cpu-baseline --no-ops --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('loss', d['config']['loss'])"; done | sort | uniq -c
