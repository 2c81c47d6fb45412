cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03prof2; mkdir -p $O; uptime
python bench.py --steps 20 --warmup 5 --cfg scannet --scenes 2 --points 100000 --no-cpu-baseline --no-ops > "$O/bench_scannet.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 --cfg scannet --scenes 2 --points 100000 --dtype bf16 --no-cpu-baseline --no-ops > "$O/bench_scannet_bf16.json" 2>/dev/null
python bench.py --steps 20 --warmup 5 --scenes 4 --points 80000 --no-cpu-baseline --no-ops > "$O/bench_4x80k.json" 2>/dev/null
for f in scannet scannet_bf16 4x80k; do python -c "import json,sys; d=json.loads(open('$O/bench_$f.json').read()); print('$f', d['ms_per_step'], d['value'])"; done; uptime
